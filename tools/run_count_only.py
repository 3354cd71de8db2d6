#!/usr/bin/env python3
"""Profiling driver: north-star (or --n) workload, graph build (+ optional sweep) a few times."""
import argparse, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import motionplanning_jl_amd as mp
ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=1000000)
ap.add_argument("--reps", type=int, default=2)
ap.add_argument("--sweep", action="store_true")
ap.add_argument("--path", type=int, default=0)
a = ap.parse_args()
w = mp.workloads.north_star(a.n)
ctx = mp.Context(0)
ctx.set_option("rdisc_path", a.path)
ctx.upload_samples(w.X); ctx.upload_boxes(w.lohi, w.ss_lo, w.ss_hi)
for _ in range(a.reps):
    nnz = ctx.graph_build_device(w.r)
    if a.sweep:
        ctx.graph_sweep_device()
print("nnz", nnz, {k: ctx.timing(k) for k in ("grid", "rdisc_count", "rdisc_fill", "rdisc_sort", "sweep_graph")}, ctx.graph_stats(), ctx.stat("survivors"))
