#!/bin/bash
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
ROOT=$(pwd)
mkdir -p gpurun_out
OUT=gpurun_out/r5_call8.txt; : > $OUT
timeout 1500 python -m pytest tests/test_gpu_wavefront.py tests/test_gpu_mirror.py tests/test_gpu_step_parity.py -x -q -k "wavefront or mirror or low_dimensional or form_grid or fmt or notebook" > gpurun_out/r5_pytest8.log 2>&1; grep -E "passed|failed|Error|assert" gpurun_out/r5_pytest8.log | tail -8 >> $OUT
cat > /tmp/wf.py <<PY
import sys, os, time
sys.path.insert(0, "$ROOT")
import motionplanning_jl_amd as mp
w = mp.workloads.north_star()
ctx = mp.Context(0)
ctx.upload_samples(w.X); ctx.upload_boxes(w.lohi, w.ss_lo, w.ss_hi)
ctx.graph_step_device(w.r)
LAZY = int(sys.argv[1])
for _ in range(3):
    t = time.perf_counter()
    res = ctx.fmtstar_wavefront(w.r, mp._lib.GOAL_BALL, w.goal_params(), band=0.25 * w.r, lazy=bool(LAZY), want_tree=False)
    print("lazy %d solve %.2f ms wavefronts %d checks %d cost %.6f" % (LAZY, 1e3 * (time.perf_counter() - t), res["info"]["iters"], res["collision_checks"], res["cost"]), flush=True)
PY
python3 /tmp/wf.py 0 >> $OUT 2>&1
python3 /tmp/wf.py 1 >> $OUT 2>&1
for L in 0; do
rm -rf /tmp/prof_wf
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_wf -o s -- python3 /tmp/wf.py $L > /tmp/wfp.log 2>&1)
DB=$(find /tmp/prof_wf -name "*_results.db" | head -1)
python3 tools/rocpd_stats.py "$DB" gpurun_out/r5_wavefront_kernel_stats_c_lazy$L.csv > /dev/null
echo "lazy $L kernel stats (name calls avg_ns)" >> $OUT
grep "k_wf\|points_free" gpurun_out/r5_wavefront_kernel_stats_c_lazy$L.csv | python3 -c "
import sys,csv
for r in csv.reader(sys.stdin):
    print('  ', r[0][:40].ljust(42), r[1], r[3])
" >> $OUT
done
for n in 400000 110000; do
timeout 300 python bench.py --workload cfg1 --n $n --no-cpu-baseline --no-cold --no-solve --steps 20 > gpurun_out/r5_bench_cfg1_n$n.json 2> /tmp/c1.err; tail -2 /tmp/c1.err >> $OUT
python - >> $OUT <<PY
import json
try:
    d = json.load(open("gpurun_out/r5_bench_cfg1_n$n.json"))
    print("cfg1 n=$n ms_per_step %.3f" % d["ms_per_step"], d["submetrics"].get("kernel_ms"), {k: d["config"].get(k) for k in ("workload", "N", "r", "nnz")}, d["submetrics"].get("forms"))
except Exception as e:
    print("cfg1 bench parse failed", e)
PY
done
timeout 900 python tools/run_form_grid.py > gpurun_out/r5_form_grid.txt 2>&1; grep "d 2" gpurun_out/r5_form_grid.txt >> $OUT
cat $OUT
