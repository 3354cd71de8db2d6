import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import motionplanning_jl_amd as mp
w = mp.workloads.north_star(int(sys.argv[1]) if len(sys.argv) > 1 else 1000000)
ctx = mp.Context(0)
ctx.upload_samples(w.X)
for pool in (0, 1):
    ctx.set_option("rdisc_pool", pool)
    ctx.timing_reset()
    for i in range(3):
        nnz = ctx.graph_build_device(w.r)
    print("pool", pool, "nnz", nnz, "pool_used", ctx.stat("pool_used"), {k: round(ctx.timing(k)[0], 3) for k in ("rdisc_count", "rdisc_fill", "rdisc_sort")}, flush=True)
