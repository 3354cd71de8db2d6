"""Per-rank step time of the sharded step, simulated on one GPU (rank g of G builds and sweeps only its shard)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import motionplanning_jl_amd as mp
w = mp.workloads.north_star()
for G in (1, 2, 4, 8):
    worst = 0.0
    for g in range(G):
        c = mp.Context(0)
        c.set_shard(g, G)
        c.upload_samples(w.X); c.upload_boxes(w.lohi, w.ss_lo, w.ss_hi)
        c.set_option("rebuild_index", 1)
        for _ in range(2):
            nnz = c.graph_step_device(w.r)
        c.timing_reset()
        torch.cuda.synchronize(); t = time.time()
        for _ in range(5):
            nnz = c.graph_step_device(w.r)
        c.graph_device_ptrs(); torch.cuda.synchronize()
        dt = (time.time() - t) / 5
        worst = max(worst, dt)
        km = {k: round(c.timing(k)[0], 3) for k in ("grid", "rdisc_count", "rdisc_sort", "sweep_graph")}
        print("G %d rank %d: nnz %d step %.3f ms %s" % (G, g, nnz, dt * 1e3, km), flush=True)
        c.close()
    print("G %d: slowest rank %.3f ms -> speedup vs G=1 needs all-gather on top" % (G, worst * 1e3), flush=True)
