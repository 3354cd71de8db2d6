#!/usr/bin/env python3
"""Which form of the edge tests a step took (stat sweep_form) and the per-step intervals, north star."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import motionplanning_jl_amd as mp
w = mp.workloads.north_star()
for form in (2, 1, 0):
    c = mp.Context(0); c.set_option("fuse_broad", form); c.set_option("rebuild_index", 1)
    c.upload_samples(w.X); c.upload_boxes(w.lohi, w.ss_lo, w.ss_hi)
    for it in range(3):
        nnz = c.graph_step_device(w.r)
        print("option", form, "step", it, "nnz", nnz, "form", c.stat("sweep_form"), "overflow", c.stat("pend_overflowed"), "half", c.stat("rdisc_half_used"), "pair items", c.stat("pair_items"), flush=True)
    c.timing_reset()
    for it in range(5): c.graph_step_device(w.r)
    print({k: round(c.timing(k)[0] * c.timing(k)[1] / 5, 3) for k in ("pair_kernel", "exact_pairs", "rdisc_sort", "sweep_graph", "sweep_kernel")}, flush=True)
    c.close()
