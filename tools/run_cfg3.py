"""BASELINE.json configs[2]: R^12 all-pairs r-disc graph, N = 1e6, E[deg] ~ 256 (brute-force regime: 2 grid cells per axis)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import motionplanning_jl_amd as mp
N = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
w = mp.workloads.cfg3(N)
c = mp.Context(0)
c.upload_samples(w.X); c.upload_boxes(w.lohi, w.ss_lo, w.ss_hi)
c.set_option("rebuild_index", 1)
for i in range(2):
    c.timing_reset()
    torch.cuda.synchronize(); t = time.time()
    nnz = c.graph_build_device(w.r); c.graph_sweep_device()
    torch.cuda.synchronize(); dt = time.time() - t
    st = c.graph_stats()
    print("N %d d 12 r %.4f nnz %d deg %.1f: step %.1f ms, pairs tested %.3g (%.3g pairs/s), path %d pool %d slices %d cells %d %s" % (
        N, w.r, nnz, nnz / N, dt * 1e3, st["pairs_tested"], st["pairs_tested"] / dt, c.stat("rdisc_path_used"), c.stat("pool_used"),
        c.stat("slices"), c.stat("cells"), {k: round(c.timing(k)[0], 2) for k in ("grid", "rdisc_count", "rdisc_fill", "rdisc_sort", "sweep_graph")}), flush=True)
