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
# steady state: count + fill + sweep repeated (buffers already allocated and touched)
import numpy as np, ctypes as C
colptr = np.empty(n + 1, dtype=np.int64); nnz = C.c_int64()
for i in range(3):
    ctx.timing_reset()
    t0 = time.time()
    ctx._chk(ctx._L.mpfmt_di_graph_count(ctx._h, float(w.rho), float(w.r), colptr.ctypes.data_as(C.POINTER(C.c_int64)), C.byref(nnz)))
    ctx.nnz = nnz.value
    ctx.di_graph_edges_free()
    print("repeat", i, {k: round(ctx.timing(k)[0], 2) for k in ("di_count", "di_fill", "di_sweep")}, "wall %.1f ms" % ((time.time() - t0) * 1e3), flush=True)
