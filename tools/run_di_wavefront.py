"""BASELINE configs[3] (kinodynamic FMT*, double integrator R^4, N = 1e5) whole solve: sequential host recursion vs the
directed wavefront form on the device.  Usage: python tools/run_di_wavefront.py [N]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import motionplanning_jl_amd as mp
N = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000
w = mp.workloads.cfg4(N)
L = mp._lib
with mp.Context(0) as c:
    c.upload_samples(w.X); c.upload_boxes(w.lohi, w.ss_lo, w.ss_hi)
    goal = w.X[-1]
    t0 = time.perf_counter(); seq = c.di_fmtstar(w.rho, w.r, L.GOAL_POINT, goal); t = time.perf_counter() - t0
    print("sequential: status %d cost %.6f checks %d  graph %.1f ms sweep %.1f ms host loop %.1f ms (total %.1f ms)" %
          (seq["status"], seq["cost"], seq["collision_checks"], seq["ms_graph"], seq["ms_sweep"], seq["ms_host_loop"], 1e3 * t))
    for bandf in (0.02, 0.05, 0.1, 0.25):
        best = None
        for rep in range(2):
            t0 = time.perf_counter()
            r = c.di_fmtstar_wavefront(w.rho, w.r, L.GOAL_POINT, goal, band=bandf * w.r, want_tree=False)
            dt = 1e3 * (time.perf_counter() - t0)
            best = dt if best is None else min(best, dt)
        print("band %.2f: status %d cost %.6f (x%.4f) %4d wavefronts %9d checks  loop %.1f ms  call %.1f ms (graph + sweep reused)" %
              (bandf, r["status"], r["cost"], r["cost"] / seq["cost"], r["info"]["iters"], r["collision_checks"], r["ms_host_loop"], best))
