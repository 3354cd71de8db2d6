#!/bin/bash
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
ROOT=$(pwd)
mkdir -p gpurun_out
OUT=gpurun_out/r5_call6.txt; : > $OUT
timeout 600 python -m pytest tests/test_gpu_wavefront.py tests/test_gpu_mirror.py -x -q 2>&1 | grep -E "passed|failed" >> $OUT
for v in default ord3 default ord3; do
  if [ $v = default ]; then unset MPFMT_LIB_PATH; else export MPFMT_LIB_PATH=$ROOT/build_ab/libmpfmt_$v.so; fi
  timeout 300 python bench.py --no-cpu-baseline --no-cold --no-solve --steps 30 > /tmp/b_$v.json 2>/dev/null
  python - >> $OUT <<PY
import json
d = json.load(open("/tmp/b_$v.json"))
k = d["submetrics"].get("kernel_ms", {})
print("$v", "ms_per_step %.3f" % d["ms_per_step"], "pair %.3f sort %.3f grid %.3f ord_per_cu %s" % (k.get("pair_kernel", 0), k.get("rdisc_sort", 0), k.get("grid", 0), d["submetrics"].get("launch", {}).get("ord_per_cu")))
PY
done
unset MPFMT_LIB_PATH
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
for L in 0 1; do
rm -rf /tmp/prof_wf
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_wf -o s -- python3 /tmp/wf.py $L > /tmp/wfp.log 2>&1)
DB=$(find /tmp/prof_wf -name "*_results.db" | head -1)
python3 tools/rocpd_stats.py "$DB" gpurun_out/r5_wavefront_kernel_stats_lazy$L.csv > /dev/null
echo "lazy $L kernel stats (name calls avg_ns)" >> $OUT
grep "k_wf\|points_free" gpurun_out/r5_wavefront_kernel_stats_lazy$L.csv | python3 -c "
import sys,csv
for r in csv.reader(sys.stdin):
    print('  ', r[0][:40].ljust(42), r[1], r[3])
" >> $OUT
done
cat $OUT
