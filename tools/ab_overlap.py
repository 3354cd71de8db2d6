import sys, os, time
sys.path.insert(0, "/root/repo")
import torch
import motionplanning_jl_amd as mp
w = mp.workloads.north_star()
for ov in (1, 0, 1, 0):
    c = mp.Context(0); c.set_option("overlap", ov); c.set_option("rebuild_index", 1)
    c.upload_samples(w.X); c.upload_boxes(w.lohi, w.ss_lo, w.ss_hi)
    for _ in range(3): c.graph_step_device(w.r)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): c.graph_step_device(w.r)
    torch.cuda.synchronize(); print("overlap", ov, "step ms", 1e3 * (time.perf_counter() - t0) / 20, "nnz", c.nnz, flush=True)
    c.close()
