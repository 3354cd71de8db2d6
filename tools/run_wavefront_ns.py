"""North-star whole solve on the device (mpfmt_fmtstar_wavefront) on the graph and mask a step left resident: per band, reading the
resident mask (the default) and testing every asked-for edge against the obstacle set (MPFMT_WF_LAZY)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import motionplanning_jl_amd as mp
w = mp.workloads.north_star()
ctx = mp.Context(0)
ctx.upload_samples(w.X); ctx.upload_boxes(w.lohi, w.ss_lo, w.ss_hi)
ctx.graph_step_device(w.r)
for lazy in (False, True):
    for bandf in (0.05, 0.25, 2.0):
        best = None
        for _ in range(3):
            t = time.perf_counter()
            res = ctx.fmtstar_wavefront(w.r, mp._lib.GOAL_BALL, w.goal_params(), band=bandf * w.r, lazy=lazy, want_tree=False)
            ms = 1e3 * (time.perf_counter() - t)
            best = ms if best is None else min(best, ms)
        print("edge tests %s band %.2f r: %.2f ms  cost %.6f  wavefronts %d  checks %d  (Group-Marching batches, not the reference's pop order)"
              % ("lazy" if lazy else "from the resident mask", bandf, best, res["cost"], res["info"]["iters"], res["collision_checks"]), flush=True)
