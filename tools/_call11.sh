cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests/test_gpu_wavefront.py tests/test_gpu_mirror.py tests/test_gpu_notebook.py -x -q 2>&1 | grep -E "passed|failed|Error|assert" | tail -5
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_boundary.py -x -q -k "fmt or wavefront or import or solve or smoke" 2>&1 | grep -E "passed|failed|Error|assert" | tail -5
python - <<'PY'
import sys, os, time
sys.path.insert(0, os.getcwd())
import motionplanning_jl_amd as mp
w = mp.workloads.north_star()
ctx = mp.Context(0)
ctx.upload_samples(w.X); ctx.upload_boxes(w.lohi, w.ss_lo, w.ss_hi)
ctx.graph_step_device(w.r)
for pos in (1, 0, 1, 0):
    ctx.set_option("wf_pos_space", pos)
    for lazy in (False, True):
        best = 1e9
        for _ in range(4):
            t = time.perf_counter()
            res = ctx.fmtstar_wavefront(w.r, mp._lib.GOAL_BALL, w.goal_params(), band=0.25 * w.r, lazy=lazy, want_tree=False)
            best = min(best, 1e3 * (time.perf_counter() - t))
        print("pos %d lazy %d solve best %.2f ms wavefronts %d checks %d cost %.6f" % (pos, lazy, best, res["info"]["iters"], res["collision_checks"], res["cost"]), flush=True)
PY
