#!/bin/bash
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
ROOT=$(pwd)
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_step_parity.py tests/test_gpu_multirank.py tests/test_gpu_north_star_step.py -x -q 2>&1 | tail -4
for ss in -1 1 0; do
python - <<PY
import sys, os, time
sys.path.insert(0, os.getcwd())
import torch
import motionplanning_jl_amd as mp
w = mp.workloads.north_star()
for G, g in ((1, 0), (8, 3), (4, 1), (2, 0)):
    c = mp.Context(0); c.set_shard(g, G); c.set_option("sort_slots", $ss)
    c.upload_samples(w.X); c.upload_boxes(w.lohi, w.ss_lo, w.ss_hi)
    c.set_option("rebuild_index", 1)
    for _ in range(3): nnz = c.graph_step_device(w.r)
    c.timing_reset(); torch.cuda.synchronize(); t = time.time()
    for _ in range(8): nnz = c.graph_step_device(w.r)
    torch.cuda.synchronize(); dt = (time.time() - t) / 8
    print("sort_slots $ss G %d rank %d: nnz %d step %.3f ms grid %.3f" % (G, g, nnz, dt * 1e3, c.timing("grid")[0]), flush=True)
    c.close()
PY
done
for spec in "3 8" "0 1"; do
  set -- $spec
  rm -rf /tmp/prof_sh
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace -d /tmp/prof_sh -o s -- python3 $ROOT/tools/run_shard_one.py $1 $2 > /tmp/sh.log 2>&1)
  DB=$(find /tmp/prof_sh -name "*_results.db" | head -1)
  python3 tools/step_timeline.py $DB 2 > gpurun_out/r5_step_timeline_b_g$2_rank$1.txt 2>&1
  cat gpurun_out/r5_step_timeline_b_g$2_rank$1.txt
done
cat > /tmp/wf.py <<PY
import sys, os, time
sys.path.insert(0, "$ROOT")
import motionplanning_jl_amd as mp
w = mp.workloads.north_star()
ctx = mp.Context(0)
ctx.upload_samples(w.X); ctx.upload_boxes(w.lohi, w.ss_lo, w.ss_hi)
ctx.graph_step_device(w.r)
for _ in range(3):
    t = time.perf_counter()
    res = ctx.fmtstar_wavefront(w.r, mp._lib.GOAL_BALL, w.goal_params(), band=0.25 * w.r, want_tree=False)
    print("solve %.2f ms wavefronts %d checks %d" % (1e3 * (time.perf_counter() - t), res["info"]["iters"], res["collision_checks"]), flush=True)
PY
rm -rf /tmp/prof_wf
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_wf -o s -- python3 /tmp/wf.py > $ROOT/gpurun_out/r5_wf.log 2>&1)
tail -5 gpurun_out/r5_wf.log
DB=$(find /tmp/prof_wf -name "*_results.db" | head -1)
python3 tools/rocpd_stats.py "$DB" gpurun_out/r5_wavefront_kernel_stats_before.csv | head -30
cat gpurun_out/r5_wavefront_kernel_stats_before.csv | head -30
