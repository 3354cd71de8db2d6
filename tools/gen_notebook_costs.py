#!/usr/bin/env python3
"""Distributions of the solve cost of the reference notebook's three set-ups through the HIP path (run on the GPU box):
tests/golden/notebook_costs.json = per set-up the seeds, the costs (null = goal not reached), the published single draw and the
straight-line bound.  usage: python tools/gen_notebook_costs.py [n_seeds]"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import motionplanning_jl_amd as mp  # noqa: E402
from motionplanning_jl_amd import notebook  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 240
out = {"generator": "tools/gen_notebook_costs.py", "reference": "docs/MotionPlanning.ipynb cells 5, 8, 11", "N": 1000, "setups": {}}
for name in ("geometric", "double_integrator", "dubins"):
    t0 = time.time()
    seeds = list(range(1000, 1000 + n))
    costs = [notebook.solve(name, s) for s in seeds]
    c = np.array([x for x in costs if x is not None])
    q = np.quantile(c, [0.01, 0.5, 0.99])
    out["setups"][name] = {"seeds": seeds, "costs": costs, "published": notebook.PUBLISHED[name], "straight_line_bound": notebook.straight_line_bound(name),
                           "solved": int(len(c)), "q01": float(q[0]), "median": float(q[1]), "q99": float(q[2])}
    print("%-18s %d/%d solved  q01 %.4f  median %.4f  q99 %.4f  published %.4f  (%.1f s)" % (name, len(c), n, q[0], q[1], q[2], notebook.PUBLISHED[name], time.time() - t0))
os.makedirs(os.path.join(ROOT, "tests", "golden"), exist_ok=True)
json.dump(out, open(os.path.join(ROOT, "tests", "golden", "notebook_costs.json"), "w"))
if os.path.isdir(os.path.join(ROOT, "gpurun_out")):
    json.dump(out, open(os.path.join(ROOT, "gpurun_out", "notebook_costs.json"), "w"))
