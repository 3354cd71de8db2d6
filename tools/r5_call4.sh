#!/bin/bash
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
ROOT=$(pwd)
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_wavefront.py tests/test_gpu_mirror.py -x -q 2>&1 | tail -4
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -k "wavefront or fmt or shard" 2>&1 | tail -3
cat > /tmp/wf.py <<PY
import sys, os, time
sys.path.insert(0, "$ROOT")
import motionplanning_jl_amd as mp
w = mp.workloads.north_star()
ctx = mp.Context(0)
ctx.upload_samples(w.X); ctx.upload_boxes(w.lohi, w.ss_lo, w.ss_hi)
ctx.graph_step_device(w.r)
for lazy in (False, True):
    for _ in range(3):
        t = time.perf_counter()
        res = ctx.fmtstar_wavefront(w.r, mp._lib.GOAL_BALL, w.goal_params(), band=0.25 * w.r, lazy=lazy, want_tree=False)
        print("lazy %d solve %.2f ms wavefronts %d checks %d cost %.6f" % (lazy, 1e3 * (time.perf_counter() - t), res["info"]["iters"], res["collision_checks"], res["cost"]), flush=True)
PY
python3 /tmp/wf.py
rm -rf /tmp/prof_wf
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_wf -o s -- python3 /tmp/wf.py > $ROOT/gpurun_out/r5_wf2.log 2>&1)
DB=$(find /tmp/prof_wf -name "*_results.db" | head -1)
python3 tools/rocpd_stats.py "$DB" gpurun_out/r5_wavefront_kernel_stats_after1.csv > /dev/null
grep "k_wf\|points_free" gpurun_out/r5_wavefront_kernel_stats_after1.csv | cut -c1-60,200-400 | sed 's/.*\(k_wf_[a-z_]*\|k_points_free\).*)",/\1 /'
for spec in "3 8" "0 1"; do
  set -- $spec
  rm -rf /tmp/prof_sh
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace -d /tmp/prof_sh -o s -- python3 $ROOT/tools/run_shard_one.py $1 $2 > /tmp/sh.log 2>&1)
  DB=$(find /tmp/prof_sh -name "*_results.db" | head -1)
  python3 tools/step_timeline.py $DB 2 > gpurun_out/r5_step_timeline_b_g$2_rank$1.txt 2>&1
  cat gpurun_out/r5_step_timeline_b_g$2_rank$1.txt
done
