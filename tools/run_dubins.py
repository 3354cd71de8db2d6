"""Dubins / Reeds-Shepp car at planner scale: graph build (positions r-disc graph + exact steering-cost filter), waypoint
sweep, plan.   usage: run_dubins.py [N] [dubins|reedsshepp]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import motionplanning_jl_amd as mp
from oracle import oracle as orc
N = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000
car = sys.argv[2] if len(sys.argv) > 2 else "dubins"
rng = np.random.default_rng(5)
X = np.column_stack([rng.random(N), rng.random(N), rng.random(N) * 2 * np.pi])
X[0] = [0.05, 0.05, 0.6]; X[-1] = [0.95, 0.95, 0.8]
lohi = mp.workloads.make_boxes(rng, 20, 2, 0.02, 0.08, [X[0, :2], X[-1, :2]])
lo, hi = np.zeros(3), np.array([1.0, 1.0, 2 * np.pi])
rt = 0.05
r = mp.workloads.fmt_radius(1.0, 3, 2 * np.pi, N)
c = mp.Context(0)
c.upload_samples(X); c.upload_boxes(lohi, lo, hi, dw=2)
for i in range(2):
    c.timing_reset()
    t = time.time()
    res = getattr(c, car + "_fmtstar")(rt, 1.0, r, mp._lib.GOAL_BALL, [0.95, 0.95, 0.05])
    print(car, "N %d r %.4f nnz %d candidates %d: status %d cost %.4f checks %d | graph %.1f ms (cost filter %.2f ms) sweep %.2f ms host %.0f ms wall %.2f s" % (
        N, r, res["nnz"], c.stat("pairs_tested"), res["status"], res["cost"], res["collision_checks"], res["ms_graph"], c.timing("car_graph")[0],
        c.timing("car_sweep")[0], res["ms_host_loop"], time.time() - t), flush=True)
P0, P1 = X[:100000 if N >= 100000 else N], X[::-1][:100000 if N >= 100000 else N]
t = time.time(); k = 0
for a, b in zip(P0[:20000], P1[:20000]):
    getattr(orc, car)(a, b, rt, 1.0); k += 1
print("oracle (1 core, through ctypes): %.3g steers/s" % (k / (time.time() - t)))
