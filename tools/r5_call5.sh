#!/bin/bash
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
ROOT=$(pwd)
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_wavefront.py tests/test_gpu_mirror.py tests/test_gpu_step_parity.py tests/test_gpu_north_star_step.py -x -q > gpurun_out/r5_pytest5.log 2>&1; grep -E "passed|failed|error" gpurun_out/r5_pytest5.log | tail -3
for v in default ord3 ord4nf; do
  if [ $v = default ]; then unset MPFMT_LIB_PATH; else export MPFMT_LIB_PATH=$ROOT/build_ab/libmpfmt_$v.so; fi
  for rep in 1 2; do
  timeout 300 python bench.py --no-cpu-baseline --no-cold --no-solve --steps 30 > /tmp/b_$v.json 2>/dev/null
  python - <<PY
import json
d = json.load(open("/tmp/b_$v.json"))
k = d["submetrics"].get("kernel_ms", {})
print("$v", "ms_per_step %.3f" % d["ms_per_step"], "pair %.3f sort %.3f grid %.3f ord_per_cu %s" % (k.get("pair_kernel", 0), k.get("rdisc_sort", 0), k.get("grid", 0), d["submetrics"].get("launch", {}).get("ord_per_cu")))
PY
  done
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
for lazy in (False, True):
    for _ in range(2):
        t = time.perf_counter()
        res = ctx.fmtstar_wavefront(w.r, mp._lib.GOAL_BALL, w.goal_params(), band=0.25 * w.r, lazy=lazy, want_tree=False)
        print("lazy %d solve %.2f ms wavefronts %d checks %d cost %.6f" % (lazy, 1e3 * (time.perf_counter() - t), res["info"]["iters"], res["collision_checks"], res["cost"]), flush=True)
PY
rm -rf /tmp/pmc_wf
(cd /tmp && timeout 600 rocprofv3 -i $ROOT/tools/pmc_mem.txt --kernel-trace --output-format csv -d /tmp/pmc_wf -o p -- python3 /tmp/wf.py > /tmp/pmc_wf.log 2>&1)
python3 tools/pmc_summary.py /tmp/pmc_wf k_wf > gpurun_out/r5_pmc_wavefront_before.txt 2>&1
cat gpurun_out/r5_pmc_wavefront_before.txt
for spec in "3 8" "0 1"; do
  set -- $spec
  rm -rf /tmp/prof_sh
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace -d /tmp/prof_sh -o s -- python3 $ROOT/tools/run_shard_one.py $1 $2 > /tmp/sh.log 2>&1)
  DB=$(find /tmp/prof_sh -name "*_results.db" | head -1)
  python3 tools/step_timeline.py $DB 2 > gpurun_out/r5_step_timeline_c_g$2_rank$1.txt 2>&1
  cat gpurun_out/r5_step_timeline_c_g$2_rank$1.txt
done
