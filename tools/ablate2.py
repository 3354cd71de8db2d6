"""Ablation of the single-pass pair kernel (results invalid while a bit is set): 1 skip sign extraction, 2 skip refine (and with it
the slot writes), 4 skip the MFMAs.  Usage: python tools/ablate2.py"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import motionplanning_jl_amd as mp
w = mp.workloads.north_star()
for ab in (0, 2, 1, 3, 7):
    ctx = mp.Context(0)
    ctx.upload_samples(w.X); ctx.upload_boxes(w.lohi, w.ss_lo, w.ss_hi)
    ctx.set_option("rebuild_index", 1)
    ctx.graph_step_device(w.r)                 # capacities learnt with the real kernel
    ctx.set_option("mf_ablate", ab)
    ctx.timing_reset()
    for i in range(4):
        try:
            ctx.graph_step_device(w.r)
        except Exception as e:
            pass
    print("ablate", ab, {k: round(ctx.timing(k)[0], 3) for k in ("rdisc_count", "rdisc_sort", "sweep_graph")}, flush=True)
    ctx.close()
