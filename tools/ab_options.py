"""A/B of ctx options on a workload's step: python tools/ab_options.py <north_star|cfg2|cfg3|cfg1:N|grid:d:N:M:deg> opt=v[,opt=v...] [...] -- every further
argument is one configuration; each is timed twice (alternating), fresh context each time."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import motionplanning_jl_amd as mp
name = sys.argv[1] if len(sys.argv) > 1 else "north_star"
cfgs = sys.argv[2:] or ["overlap=1", "overlap=0"]
if name.startswith("grid:"):          # grid:d:N:M:deg -- a world of tools/run_form_grid.py (unit cube, M boxes, radius for a mean degree)
    import numpy as np
    d_, N_, M_, deg_ = (float(v) for v in name.split(":")[1:])
    rng = np.random.default_rng(5)
    class W: pass
    w = W(); w.X = rng.random((int(N_), int(d_))); w.r = float((deg_ / N_) ** (1.0 / d_) * 0.62)
    w.lohi = mp.workloads.make_boxes(rng, int(M_), int(d_), 0.05, 0.25, []); w.ss_lo = np.zeros(int(d_)); w.ss_hi = np.ones(int(d_))
else:
    w = mp.workloads.cfg1(N=int(name.split(":")[1])) if name.startswith("cfg1:") else mp.workloads.BY_NAME[name]()
steps = 10 if name == "cfg3" else 30
for cfg in cfgs * 2:
    c = mp.Context(0); c.set_option("rebuild_index", 1)
    for kv in cfg.split(","):
        k, v = kv.split("="); c.set_option(k, int(v))
    c.upload_samples(w.X); c.upload_boxes(w.lohi, w.ss_lo, w.ss_hi)
    for _ in range(3): c.graph_step_device(w.r)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(steps): c.graph_step_device(w.r)
    torch.cuda.synchronize(); print(name, cfg, "step ms %.4f" % (1e3 * (time.perf_counter() - t0) / steps), "nnz", c.nnz, flush=True)
    c.close()
