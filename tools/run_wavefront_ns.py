"""North-star whole solve with the recursion on the device (mpfmt_fmtstar_wavefront): band sweep, cost vs the sequential
recursion, wall time.  Usage: python tools/run_wavefront_ns.py [N]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import motionplanning_jl_amd as mp
N = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
w = mp.workloads.north_star(N)
L = mp._lib
with mp.Context(0) as c:
    c.upload_samples(w.X); c.upload_boxes(w.lohi, w.ss_lo, w.ss_hi)
    t0 = time.perf_counter(); seq = c.fmtstar(w.r, L.GOAL_BALL, w.goal_params()); t_seq = time.perf_counter() - t0
    print("sequential (host recursion): cost %.6f checks %d  total %.1f ms (host loop %.1f ms)" % (seq["cost"], seq["collision_checks"], 1e3 * t_seq, seq["ms_host_loop"]))
    for bandf in (0.0, 0.05, 0.1, 0.25, 0.5, 1.0, 2.0):
        best = None
        for rep in range(3):
            t0 = time.perf_counter()
            r = c.fmtstar_wavefront(w.r, L.GOAL_BALL, w.goal_params(), band=bandf * w.r, want_tree=False)
            dt = 1e3 * (time.perf_counter() - t0)
            best = dt if best is None else min(best, dt)
        print("band %.2f r: cost %.6f (x%.5f)  %4d wavefronts  %8d checks  %7d connected  %.2f ms" %
              (bandf, r["cost"], r["cost"] / seq["cost"], r["info"]["iters"], r["collision_checks"], r["info"]["tot_conn"], best))
