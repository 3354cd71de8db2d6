"""Closest obstacle points at scale: n points x M boxes in R^d with a random SPD weight.
usage: run_closest.py [n] [d] [M]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import motionplanning_jl_amd as mp
from oracle import oracle as orc
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000
d = int(sys.argv[2]) if len(sys.argv) > 2 else 6
M = int(sys.argv[3]) if len(sys.argv) > 3 else 200
rng = np.random.default_rng(9)
c = rng.random((M, d)); h = 0.02 + 0.1 * rng.random((M, d))
lohi = np.stack([c - h, c + h], axis=1)
P = rng.random((n, d))
A = rng.standard_normal((d, d)); W = A @ A.T + 0.3 * np.eye(d)
ctx = mp.Context(0)
ctx.upload_samples(P[:2]); ctx.upload_boxes(lohi, None, None)
for i in range(3):
    ctx.timing_reset()
    t = time.time()
    d2, v, k, fails = ctx.closest(P, W)
    wall = time.time() - t
    tp, ts = ctx.timing("closest_pairs")[0], ctx.timing("closest_select")[0]
    print("closest n %d d %d M %d: pairs kernel %.3f ms = %.3g bvls solves/s, select %.3f ms, wall %.1f ms, failures %d (%.2f%% of pairs)" % (
        n, d, M, tp, n * M / (tp * 1e-3), ts, wall * 1e3, fails, 100.0 * fails / (n * M)), flush=True)
r2 = float(np.quantile(d2, 0.5)) * 2
for i in range(2):
    ctx.timing_reset()
    t = time.time()
    ptr, idx, dd, vv, _ = ctx.closeR(P, W, r2)
    print("closeR r2 %.3g: %d entries (%.1f per point), pairs %.3f ms, fill %.3f ms, wall %.1f ms" % (
        r2, len(idx), len(idx) / n, ctx.timing("closest_pairs")[0], ctx.timing("closest_select")[0], (time.time() - t) * 1e3), flush=True)
m = min(n, 300)
t = time.time()
od2, ov, ok, obad = orc.closest_boxes(P[:m], lohi, W)
dt = time.time() - t
print("oracle (1 core): %.3g bvls solves/s; agreement on %d points: max |d2 - d2_oracle| %.2e, same box %.4f" % (
    m * M / dt, m, np.abs(d2[:m] - od2).max(), (k[:m] - 1 == ok).mean()))
