#!/usr/bin/env python3
"""North-star solve on the resident graph of a step: lazy edge tests against the swept mask (MPFMT_WF_EAGER)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import motionplanning_jl_amd as mp
w = mp.workloads.north_star()
c = mp.Context(0); c.upload_samples(w.X); c.upload_boxes(w.lohi, w.ss_lo, w.ss_hi)
c.graph_step_device(w.r); c.graph_step_device(w.r)
import numpy as np
for eager in (False, True):
    for band in (0.25, 2.0):
        ts = []
        for it in range(4):
            t0 = time.perf_counter()
            res = c.fmtstar_wavefront(w.r, mp._lib.GOAL_BALL, w.goal_params(), band=band * w.r, eager=eager, want_tree=False)
            ts.append(1e3 * (time.perf_counter() - t0))
        print("eager", eager, "band %.2f r: %.2f ms  cost %.6f  wavefronts %d checks %d  (Group-Marching batches, not the reference's pop order)" % (band, min(ts), res["cost"], res["info"]["iters"], res["collision_checks"]), flush=True)
