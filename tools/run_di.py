#!/usr/bin/env python3
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import motionplanning_jl_amd as mp
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
w = mp.workloads.cfg4(n)
ctx = mp.Context(0)
ctx.upload_samples(w.X); ctx.upload_boxes(w.lohi, w.ss_lo, w.ss_hi)
t0 = time.time()
res = ctx.di_fmtstar(w.rho, w.r, mp._lib.GOAL_POINT, w.X[-1])
dt = time.time() - t0
print("N", n, "status", res["status"], "cost", res["cost"], "nnz", res["nnz"], "checks", res["collision_checks"],
      "ms_graph", res["ms_graph"], "ms_sweep", res["ms_sweep"], "ms_host", res["ms_host_loop"], "wall", dt)
print({k: ctx.timing(k) for k in ("di_count", "di_fill", "di_sweep")}, ctx.stat("pairs_tested"), ctx.stat("survivors"))
