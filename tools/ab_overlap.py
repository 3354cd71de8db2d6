"""A/B of ctx options on the north-star step: python tools/ab_overlap.py opt=v[,opt=v...] [opt=v...] -- every argument is one configuration"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import motionplanning_jl_amd as mp
cfgs = sys.argv[1:] or ["overlap=1", "overlap=0"]
w = mp.workloads.north_star()
for cfg in cfgs * 2:
    c = mp.Context(0); c.set_option("rebuild_index", 1)
    for kv in cfg.split(","):
        k, v = kv.split("="); c.set_option(k, int(v))
    c.upload_samples(w.X); c.upload_boxes(w.lohi, w.ss_lo, w.ss_hi)
    for _ in range(3): c.graph_step_device(w.r)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): c.graph_step_device(w.r)
    torch.cuda.synchronize(); print(cfg, "step ms %.4f" % (1e3 * (time.perf_counter() - t0) / 20), "nnz", c.nnz, flush=True)
    c.close()
