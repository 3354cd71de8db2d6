#!/bin/bash
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
ROOT=$(pwd)
mkdir -p gpurun_out
OUT=gpurun_out/r5_call7.txt; : > $OUT
timeout 1500 python -m pytest tests/test_gpu_wavefront.py tests/test_gpu_mirror.py tests/test_gpu_step_parity.py tests/test_gpu_stream.py -x -q > gpurun_out/r5_pytest7.log 2>&1; grep -E "passed|failed|Error|assert" gpurun_out/r5_pytest7.log | tail -8 >> $OUT
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
for L in 0; do
rm -rf /tmp/prof_wf
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_wf -o s -- python3 /tmp/wf.py $L > /tmp/wfp.log 2>&1)
DB=$(find /tmp/prof_wf -name "*_results.db" | head -1)
python3 tools/rocpd_stats.py "$DB" gpurun_out/r5_wavefront_kernel_stats_b_lazy$L.csv > /dev/null
echo "lazy $L kernel stats (name calls avg_ns)" >> $OUT
grep "k_wf\|points_free" gpurun_out/r5_wavefront_kernel_stats_b_lazy$L.csv | python3 -c "
import sys,csv
for r in csv.reader(sys.stdin):
    print('  ', r[0][:40].ljust(42), r[1], r[3])
" >> $OUT
done
timeout 600 python bench.py --workload cfg3 --no-cpu-baseline --no-cold --no-solve --steps 5 --warmup 2 > gpurun_out/r5_bench_cfg3_a.json 2> /tmp/cfg3.err; tail -2 /tmp/cfg3.err >> $OUT
python - >> $OUT <<PY
import json
try:
    d = json.load(open("gpurun_out/r5_bench_cfg3_a.json"))
    print("cfg3 ms_per_step %.2f" % d["ms_per_step"], d["submetrics"].get("kernel_ms"), d["config"])
except Exception as e:
    print("cfg3 bench parse failed", e)
PY
timeout 300 python bench.py --no-cpu-baseline --no-cold --steps 30 > gpurun_out/r5_bench_ns_a.json 2>/dev/null
python - >> $OUT <<PY
import json
d = json.load(open("gpurun_out/r5_bench_ns_a.json"))
print("ns ms_per_step %.3f" % d["ms_per_step"], d["submetrics"].get("kernel_ms"), "solve", {k: v for k, v in d["submetrics"].get("fmt_solve", {}).items() if k.startswith("ms")})
PY
cat $OUT
