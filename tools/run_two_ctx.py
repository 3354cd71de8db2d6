#!/usr/bin/env python3
"""How much of one step's time is latency that a second, independent step could fill?  K contexts on one GPU, each on its own stream,
each running the whole north-star step; wall time per round of K steps against K x the single-context step."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import motionplanning_jl_amd as mp
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
w = mp.workloads.north_star() if hasattr(mp.workloads, "north_star") else mp.workloads.cfg2(1000000)
for K in (1, 2, 3):
    ctxs, streams = [], []
    for k in range(K):
        c = mp.Context(0); s = torch.cuda.Stream(dev)
        c.set_stream(s.cuda_stream); c.upload_samples(w.X); c.upload_boxes(w.lohi, w.ss_lo, w.ss_hi)
        ctxs.append(c); streams.append(s)
    for it in range(3):
        for c in ctxs: c.graph_step_launch(w.r)
        for c in ctxs: c.graph_step_finish()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    R = 15
    for it in range(R):
        for c in ctxs: c.graph_step_launch(w.r)
        for c in ctxs: c.graph_step_finish()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / R * 1e3
    print("contexts %d: %.3f ms per round, %.3f ms per step" % (K, dt, dt / K), flush=True)
    del ctxs
