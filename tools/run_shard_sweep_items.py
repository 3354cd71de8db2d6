"""Per-rank step of rank g of G for several mf_target_items (work items the MFMA pair kernel aims for)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import motionplanning_jl_amd as mp
w = mp.workloads.north_star()
for G, g in ((8, 4), (4, 2), (2, 1), (1, 0)):
    for items in (12000, 20000, 30000, 40000, 70000):
        c = mp.Context(0)
        c.set_shard(g, G)
        c.upload_samples(w.X); c.upload_boxes(w.lohi, w.ss_lo, w.ss_hi)
        c.set_option("rebuild_index", 1); c.set_option("mf_target_items", items)
        for _ in range(3):
            c.graph_step_device(w.r)
        c.timing_reset()
        torch.cuda.synchronize(); t = time.time()
        for _ in range(6):
            c.graph_step_device(w.r)
        torch.cuda.synchronize()
        dt = (time.time() - t) / 6 * 1e3
        print("G %d rank %d items %6d: slices %2d step %.3f ms rdisc %.3f sort %.3f  pairs %.3g (%.3g/s) nnz %d" % (G, g, items, c.stat("slices"), dt, c.timing("rdisc_count")[0], c.timing("rdisc_sort")[0], c.stat("pairs_tested"), c.stat("pairs_tested") / (c.timing("rdisc_count")[0] * 1e-3), c.nnz), flush=True)
        del c
