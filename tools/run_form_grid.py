"""Which form of the edge tests a step ends up with over a grid of worlds (unit-cube state space, default options): prints
(half build used, edge-test form, pending-list overflow, pair items) for three consecutive steps of each.
Usage: python tools/run_form_grid.py"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import motionplanning_jl_amd as mp

rng = np.random.default_rng(5)
for d in (2, 3, 4, 6):
    for N in (20000, 110000, 400000):
        for M in (30, 256):
            for deg in (6, 60):
                X = rng.random((N, d)); r = float((deg / N) ** (1.0 / d) * 0.62)
                lohi = mp.workloads.make_boxes(rng, M, d, 0.05, 0.25, [])
                with mp.Context(0) as c:
                    c.set_option("rebuild_index", 1)
                    out = []
                    for it in range(3):
                        c.upload_samples(X); c.upload_boxes(lohi, np.zeros(d), np.ones(d))
                        nnz = c.graph_step_device(r)
                        out.append((c.stat("rdisc_path_used"), c.stat("rdisc_half_used"), c.stat("sweep_form"), c.stat("pend_overflowed"), c.stat("pair_items")))
                print("d %d N %6d M %3d deg %2d nnz %9d: (path, half, form, overflow, items) %s" % (d, N, M, deg, nnz, out), flush=True)
