"""Throughput of the 2-D SAT world sweep beside the scalar oracle."""
import sys, os, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import motionplanning_jl_amd as mp
from oracle import oracle as orc
fx = json.load(open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "shapes_2d.json")))
shapes = [("circle", tuple(s[1]), s[2]) if s[0] == "circle" else ("polygon", [tuple(p) for p in s[1]]) for s in fx["worlds"]["ISRR_POLY_WITH_SPIKE"]]
c = mp.Context(0)
N = 1_000_000
c.upload_shapes2d(shapes, np.zeros(2), np.ones(2))
X, att = c.sample_free(3, N, init=[0.05, 0.05], goal_kind=mp._lib.GOAL_BALL, goal_params=[0.95, 0.95, 0.03], goal_ct=1)
r = mp.workloads.fmt_radius(1.5, 2, 1.0, N)
for i in range(3):
    c.timing_reset()
    nnz = c.graph_build_device(r)
    c.graph_sweep_device()
    tg = c.timing("rdisc_count")[0] + c.timing("rdisc_sort")[0] + c.timing("grid")[0]
    ts = c.timing("sweep_graph")[0]
    print("N %d r %.5f nnz %d: graph %.2f ms, SAT sweep %.3f ms -> %.3g edges/s" % (N, r, nnz, tg, ts, nnz / ts * 1e3), flush=True)
res = c.fmtstar(r, mp._lib.GOAL_BALL, [0.95, 0.95, 0.03])
print("plan: status %d cost %.4f checks %d host loop %.0f ms" % (res["status"], res["cost"], res["collision_checks"], res["ms_host_loop"]))
S = orc.Shapes2D(shapes)
P, Q = X[:200000], X[200000:400000]
t = time.time(); m = orc.motions_free_2d(P, Q, S, np.zeros(2), np.ones(2)); dt = time.time() - t
print("oracle (1 core): %.3g segment checks/s" % (len(P) / dt))
