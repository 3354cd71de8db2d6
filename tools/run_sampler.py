"""Throughput of the batch free-space sampler at the north-star size, beside the scalar loop (oracle) on a sample."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import motionplanning_jl_amd as mp
from oracle import oracle as orc
w = mp.workloads.north_star()
c = mp.Context(0)
c.upload_boxes(w.lohi, w.ss_lo, w.ss_hi)
for i in range(3):
    c.timing_reset()
    t = time.time()
    X, att = c.sample_free(10 + i, w.N, init=w.init, goal_kind=mp._lib.GOAL_BALL, goal_params=w.goal_params(), goal_ct=5)
    dt = time.time() - t
    print("device sampler: N %d attempts %d wall %.1f ms (device loop %.2f ms) -> %.3g samples/s" % (w.N, att, dt * 1e3, c.timing("sample_free")[0], w.N / dt), flush=True)
t = time.time()
rc, W, oatt = orc.sample_free(10, 50000, 6, w.init, w.lohi, w.ss_lo, w.ss_hi, mp._lib.GOAL_BALL, w.goal_params(), goal_ct=5)
dt = time.time() - t
print("scalar loop (oracle, 1 core): N 50000 attempts %d %.1f ms -> %.3g samples/s" % (oatt, dt * 1e3, 50000 / dt))
