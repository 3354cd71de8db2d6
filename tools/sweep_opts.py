#!/usr/bin/env python3
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import motionplanning_jl_amd as mp
w = mp.workloads.north_star()
ctx = mp.Context(0)
ctx.upload_samples(w.X); ctx.upload_boxes(w.lohi, w.ss_lo, w.ss_hi)
for mode in (128, 512, 2048):
  for tgt in (40000, 70000, 120000, 160000, 250000):
    ctx.set_option("mf_xcd_mode", mode)
    ctx.set_option("mf_target_items", tgt)
    ctx.timing_reset()
    for _ in range(4):
        ctx.graph_build_device(w.r)
    print("xcd_mode", mode, "target", tgt, "slices", ctx.stat("slices"), "pool", ctx.stat("pool_used"), {k: round(ctx.timing(k)[0], 3) for k in ("rdisc_count", "rdisc_fill", "rdisc_sort")}, flush=True)
