import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import motionplanning_jl_amd as mp
w = mp.workloads.north_star()
ctx = mp.Context(0)
ctx.upload_samples(w.X)
ctx.set_option("rdisc_pool", 0)
for ab in (0, 4, 8):
    ctx.set_option("mf_ablate", ab)
    ctx.timing_reset()
    for i in range(3):
        try:
            ctx.graph_build_device(w.r)
        except Exception as e:
            pass
    print("ablate", ab, {k: round(ctx.timing(k)[0], 3) for k in ("rdisc_count",)}, flush=True)
