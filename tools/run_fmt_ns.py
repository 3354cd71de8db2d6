"""Whole FMT* solve at the north-star size: device phases vs the sequential host recursion."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import motionplanning_jl_amd as mp
w = mp.workloads.north_star()
c = mp.Context(0)
c.upload_samples(w.X); c.upload_boxes(w.lohi, w.ss_lo, w.ss_hi)
for i in range(2):
    t = time.time()
    res = c.fmtstar(w.r, mp._lib.GOAL_BALL, w.goal_params())
    print("status %d cost %.6f checks %d path %d nnz %d | graph %.1f ms sweep %.1f ms host loop %.0f ms wall %.2f s" % (
        res["status"], res["cost"], res["collision_checks"], len(res["path"]), res["nnz"], res["ms_graph"], res["ms_sweep"],
        res["ms_host_loop"], time.time() - t), flush=True)
