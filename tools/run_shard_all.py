"""Per-rank step time of every rank of a G-shard run, simulated on one GPU.  usage: run_shard_all.py G"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import motionplanning_jl_amd as mp
G = int(sys.argv[1]) if len(sys.argv) > 1 else 8
w = mp.workloads.north_star()
tot = 0
for g in range(G):
    c = mp.Context(0)
    c.set_shard(g, G)
    c.upload_samples(w.X); c.upload_boxes(w.lohi, w.ss_lo, w.ss_hi)
    c.set_option("rebuild_index", 1)
    for _ in range(3):
        c.graph_step_device(w.r)
    c.timing_reset()
    torch.cuda.synchronize(); t = time.time()
    for _ in range(6):
        nnz = c.graph_step_device(w.r)
    torch.cuda.synchronize()
    dt = (time.time() - t) / 6 * 1e3
    a, b, _ = c.shard_info()
    tot += nnz
    print("G %d rank %d: columns %7d nnz %9d pairs %.3g step %.3f ms  rdisc %.3f sort %.3f sweep %.3f" % (
        G, g, b - a, nnz, c.stat("pairs_tested"), dt, c.timing("rdisc_count")[0], c.timing("rdisc_sort")[0], c.timing("sweep_graph")[0]), flush=True)
    c.close()
print("total nnz", tot)
