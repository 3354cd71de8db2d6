#!/bin/bash
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
mkdir -p gpurun_out
python - <<'PY' > gpurun_out/r5_shard_pairs.txt 2>&1
import sys, os
sys.path.insert(0, os.getcwd())
import motionplanning_jl_amd as mp
w = mp.workloads.north_star()
for G in (1, 8):
    for g in range(G):
        c = mp.Context(0); c.set_shard(g, G)
        c.upload_samples(w.X); c.upload_boxes(w.lohi, w.ss_lo, w.ss_hi)
        c.set_option("rebuild_index", 1)
        for _ in range(2): nnz = c.graph_step_device(w.r)
        print("G", G, "rank", g, "nnz", nnz, "stats", c.graph_stats(), "slices", c.stat("slices"), "list_cap", c.stat("list_cap"), "survivors", c.stat("survivors"), flush=True)
        c.close()
PY
cat gpurun_out/r5_shard_pairs.txt
for spec in "3 8" "0 8" "0 1"; do
  set -- $spec
  rm -rf /tmp/prof_sh
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace -d /tmp/prof_sh -o s -- python3 $OLDPWD/tools/run_shard_one.py $1 $2 > /tmp/sh.log 2>&1)
  DB=$(find /tmp/prof_sh -name "*_results.db" | head -1)
  python3 tools/step_timeline.py $DB 2 > gpurun_out/r5_step_timeline_g$2_rank$1.txt 2>&1
  cat gpurun_out/r5_step_timeline_g$2_rank$1.txt
done
cat > /tmp/wf.py <<'PY'
import sys, os, time
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
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
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_wf -o s -- python3 /tmp/wf.py > gpurun_out/r5_wf.log 2>&1)
cat gpurun_out/r5_wf.log | tail -5
DB=$(find /tmp/prof_wf -name "*_results.db" | head -1)
python3 tools/rocpd_stats.py "$DB" gpurun_out/r5_wavefront_kernel_stats.csv | head -30
timeout 600 python -m pytest tests/test_gpu_boundary.py -x -q 2>&1 | tail -3
