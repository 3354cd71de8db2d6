"""BASELINE.json configs[4]: Monte-Carlo collision probability, 1e6 rollouts per candidate edge (R^6, 200 boxes)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import motionplanning_jl_amd as mp
from oracle import oracle as orc
w = mp.workloads.cfg2(20000)
c = mp.Context(0)
c.upload_samples(w.X); c.upload_boxes(w.lohi, w.ss_lo, w.ss_hi)
colptr, rowval, _ = c.rdisc_graph(w.r)
rng = np.random.default_rng(0)
pick = rng.choice(len(rowval), 256, replace=False)
cols = np.repeat(np.arange(1, w.N + 1), np.diff(colptr))
src, dst = rowval[pick], cols[pick]
for E, R in ((1, 1_000_000), (256, 1_000_000)):
    c.mc_edges_collision(src[:E], dst[:E], 0.02, 1000, seed=1)
    c.timing_reset()
    t = time.time()
    hits = c.mc_edges_collision(src[:E], dst[:E], 0.02, R, seed=1)
    dt = time.time() - t
    print("E %d x %d rollouts: kernel %.3f ms (wall %.1f ms) -> %.3g rollouts/s; mean P(collision) %.4f" % (
        E, R, c.timing("mc_edges")[0], dt * 1e3, E * R / (c.timing("mc_edges")[0] * 1e-3), hits.mean() / R), flush=True)
t = time.time(); h = orc.mc_edges(w.X, src[:1] - 1, dst[:1] - 1, 0.02, 200000, 1, w.lohi, w.ss_lo, w.ss_hi); dt = time.time() - t
print("scalar loop (1 core): %.3g rollouts/s" % (200000 / dt))
