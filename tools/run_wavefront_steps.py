"""Two device FMT* solves on the resident north-star graph (band 0.25 r), then the same solve step by step for the batch / candidate counts:
the run tools/wavefront_steps.py reads the kernel trace of."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import motionplanning_jl_amd as mp
w = mp.workloads.north_star()
ctx = mp.Context(0)
ctx.upload_samples(w.X); ctx.upload_boxes(w.lohi, w.ss_lo, w.ss_hi)
ctx.graph_step_device(w.r)
for _ in range(2):
    res = ctx.fmtstar_wavefront(w.r, mp._lib.GOAL_BALL, w.goal_params(), band=0.25 * w.r, want_tree=False)
ctx.wf_begin(w.r, mp._lib.GOAL_BALL, w.goal_params(), band=0.25 * w.r)
rows = []
while True:
    info = ctx.wf_step()
    rows.append((info["nz"], info["nx"], info["nconn"]))
    if info["done"]: break
print("(batch nodes, candidates, connected) per wavefront:", rows)
