"""One line per BASELINE.json configuration that is a whole plan (cfg1, cfg2): device phases and host recursion."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import motionplanning_jl_amd as mp
for name in ("cfg1", "cfg2"):
    w = mp.workloads.BY_NAME[name]()
    c = mp.Context(0)
    c.upload_samples(w.X); c.upload_boxes(w.lohi, w.ss_lo, w.ss_hi)
    for i in range(3):
        c.upload_samples(w.X)
        t = time.time()
        res = c.fmtstar(w.r, mp._lib.GOAL_BALL, w.goal_params())
        wall = time.time() - t
    print("%s N %d d %d M %d r %.4f: status %d cost %.6f checks %d nnz %d | graph %.2f ms sweep %.2f ms host loop %.2f ms wall %.1f ms" % (
        name, w.N, w.d, w.M, w.r, res["status"], res["cost"], res["collision_checks"], res["nnz"], res["ms_graph"], res["ms_sweep"],
        res["ms_host_loop"], wall * 1e3), flush=True)
    c.close()
