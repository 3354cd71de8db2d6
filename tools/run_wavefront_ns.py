"""North-star whole solve on the device (mpfmt_fmtstar_wavefront) with and without the captured hipGraph of a step group, per band."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import motionplanning_jl_amd as mp
w = mp.workloads.north_star()
ctx = mp.Context(0)
ctx.upload_samples(w.X); ctx.upload_boxes(w.lohi, w.ss_lo, w.ss_hi)
ctx.graph_step_device(w.r)
for graphs in (0, 1):
    ctx.set_option("wf_graphs", graphs)
    for bandf in (0.05, 0.25, 2.0):
        best = None
        for _ in range(4):
            t = time.perf_counter()
            res = ctx.fmtstar_wavefront(w.r, mp._lib.GOAL_BALL, w.goal_params(), band=bandf * w.r, want_tree=False)
            ms = 1e3 * (time.perf_counter() - t)
            best = ms if best is None else min(best, ms)
        print("graphs %d band %.2f r: %.2f ms  cost %.6f  wavefronts %d  checks %d  (Group-Marching batches, not the reference's pop order)"
              % (graphs, bandf, best, res["cost"], res["info"]["iters"], res["collision_checks"]))
