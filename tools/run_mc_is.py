#!/usr/bin/env python3
"""Importance-sampling estimator of the edge collision probability (mpfmt_mc_edges_collision_is) on the north-star world: throughput,
and the variance of the estimate against plain Monte Carlo at the same number of rollouts, for graph edges whose probability is small."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import motionplanning_jl_amd as mp
w = mp.workloads.north_star(20000)
c = mp.Context(0); c.upload_samples(w.X); c.upload_boxes(w.lohi, w.ss_lo, w.ss_hi)
colptr, rowval, _ = c.rdisc_graph(w.r * 1.6)
free = mp._lib.unpack_bits(c.graph_edges_free(), len(rowval))
cols = np.repeat(np.arange(1, w.N + 1), np.diff(colptr))
rng = np.random.default_rng(5)
pick = rng.choice(np.flatnonzero(free), 256, replace=False)
src, dst = rowval[pick].astype(np.int64), cols[pick].astype(np.int64)
sigma = 0.02
for name, f in (("plain", lambda n, s: c.mc_edges_collision(src, dst, sigma, n, seed=s)), ("importance", lambda n, s: c.mc_edges_collision_is(src, dst, sigma, n, seed=s)[0]),
                ("adaptive", lambda n, s: c.mc_edges_collision_ais(src, dst, sigma, n, seed=s)[0])):
    f(1000, 0)
    t0 = time.perf_counter(); f(1_000_000, 1); dt = time.perf_counter() - t0
    print("%-10s 256 edges x 1e6 rollouts: %.1f ms = %.2e rollouts/s" % (name, 1e3 * dt, 256e6 / dt), flush=True)
n, S = 100_000, 24
mc = np.array([c.mc_edges_collision(src, dst, sigma, n, seed=s) / n for s in range(S)])
isv = np.array([c.mc_edges_collision_is(src, dst, sigma, n, seed=s)[0] for s in range(S)])
ais = np.array([c.mc_edges_collision_ais(src, dst, sigma, n, seed=s)[0] for s in range(S)])
ref = np.mean([c.mc_edges_collision(src, dst, sigma, 1_000_000, seed=1000 + s) / 1e6 for s in range(8)], axis=0)
for lo, hi in ((1e-6, 1e-4), (1e-4, 1e-2), (1e-2, 1.0)):
    sel = (ref >= lo) & (ref < hi)
    if sel.sum() == 0:
        continue
    vr = mc[:, sel].var(axis=0, ddof=1) / np.maximum(isv[:, sel].var(axis=0, ddof=1), 1e-300)
    bias = np.abs(isv[:, sel].mean(axis=0) - ref[sel]) / np.sqrt(isv[:, sel].var(axis=0, ddof=1) / S + ref[sel] / 8e6)
    print("edges with p in [%g, %g): %3d   variance(plain) / variance(importance): median %.1f, quartiles %.1f .. %.1f   |mean - reference| / s.e.: max %.1f"
          % (lo, hi, sel.sum(), np.median(vr), np.percentile(vr, 25), np.percentile(vr, 75), bias.max()), flush=True)
    va = mc[:, sel].var(axis=0, ddof=1) / np.maximum(ais[:, sel].var(axis=0, ddof=1), 1e-300)
    ba = np.abs(ais[:, sel].mean(axis=0) - ref[sel]) / np.sqrt(ais[:, sel].var(axis=0, ddof=1) / S + ref[sel] / 8e6)
    print("                                  variance(plain) / variance(ADAPTIVE)  : median %.1f, quartiles %.1f .. %.1f   |mean - reference| / s.e.: max %.1f"
          % (np.median(va), np.percentile(va, 25), np.percentile(va, 75), ba.max()), flush=True)
print("edges with p < 1e-6 (plain MC sees nothing in 8e6 rollouts):", int((ref < 1e-6).sum()), " importance estimates there: median %.2e" % np.median(isv[:, ref < 1e-6].mean(axis=0)) if (ref < 1e-6).any() else "")
