"""Per-rank step of the sharded north star on one GPU under the two cell orders (block-major / row-major shards) and with / without the
shard + halo index: step time, pairs tested by the filter, exact hits, records the rank's columns hold."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import motionplanning_jl_amd as mp
w = mp.workloads.north_star()
G = int(sys.argv[1]) if len(sys.argv) > 1 else 8
for blocks, halo in ((1, 1), (0, 1), (1, 0), (0, 0)):
    worst, pairs, surv = 0.0, 0, 0
    for g in range(G):
        c = mp.Context(0)
        c.set_shard(g, G); c.set_option("shard_blocks", blocks); c.set_option("index_halo", halo)
        c.upload_samples(w.X); c.upload_boxes(w.lohi, w.ss_lo, w.ss_hi)
        c.set_option("rebuild_index", 1)
        for _ in range(3): nnz = c.graph_step_device(w.r)
        c.timing_reset(); torch.cuda.synchronize(); t = time.time()
        for _ in range(6): nnz = c.graph_step_device(w.r)
        torch.cuda.synchronize(); dt = (time.time() - t) / 6
        worst = max(worst, dt); st = c.graph_stats(); pairs += st["pairs_tested"]; surv += c.stat("survivors")
        print("  blocks %d halo %d G %d rank %d: step %.3f ms grid %.3f count %.3f sort %.3f  pairs %.3e survivors %.3e nnz %d"
              % (blocks, halo, G, g, dt * 1e3, c.timing("grid")[0], c.timing("rdisc_count")[0], c.timing("rdisc_sort")[0], st["pairs_tested"], c.stat("survivors"), nnz), flush=True)
        c.close()
    print("blocks %d halo %d G %d: slowest rank %.3f ms, sum of pairs tested %.4e, sum of survivors %.4e" % (blocks, halo, G, worst * 1e3, pairs, surv), flush=True)
