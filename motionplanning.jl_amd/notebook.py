"""The three set-ups of the reference's notebook (docs/MotionPlanning.ipynb cells 4-5, 7-8, 10-11) through the HIP path, as pure
functions of a seed: obstacle set ISRR_POLY_WITH_SPIKE (test/obstaclesets/2D.jl), N = 1000,

  geometric         UnitHypercube(2), init (.1, .1), PointGoal((.9, .9)), fmtstar!(P, 1000, rm = 1.5)              published 1.2346
  double_integrator DoubleIntegrator(2, vmax = .5), init (.1, .1, 0, 0), StateGoal((.9, .9, 0, 0)), r = 1.          published 5.7236
  dubins            DubinsQuasiMetricSpace(.15), init SE2State(.1, .1, 0), PointGoal((.9, .9)), r = .3, ensure_goal_ct = 10
                                                                                                                     published 1.7723
The reference samples from Julia's unseeded global RNG, so its three numbers are single draws from the distributions this module
samples; tools/gen_notebook_costs.py records those distributions, tests/test_gpu_notebook.py checks the published draws lie
inside them."""
import json
import math
import os

import numpy as np

from . import mirror as mp

PUBLISHED = {"geometric": 1.234603722643713, "double_integrator": 5.7235789452110915, "dubins": 1.7723017293799017}
_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_WORLD = None


def world():
    global _WORLD
    if _WORLD is None:
        fx = json.load(open(os.path.join(_ROOT, "tests", "golden", "shapes_2d.json")))["worlds"]["ISRR_POLY_WITH_SPIKE"]
        _WORLD = [mp.Circle(s[1], s[2]) if s[0] == "circle" else mp.Polygon(s[1]) for s in fx]
    return _WORLD


def straight_line_bound(name):
    """What no solution can beat: the Euclidean length of init -> goal; for the double integrator the cost is time + control effort
    at |v| <= vmax = 0.5, so the time alone is at least length / vmax."""
    length = math.hypot(0.8, 0.8)
    return length / 0.5 if name == "double_integrator" else length


def problem(name, ctx=None):
    CC = mp.PointRobot2D(mp.Compound2D(world()))
    if name == "geometric":
        return mp.MPProblem(mp.UnitHypercube(2), [0.1, 0.1], mp.PointGoal([0.9, 0.9]), CC, ctx), dict(rm=1.5)
    if name == "double_integrator":
        return mp.MPProblem(mp.DoubleIntegrator(2, vmax=0.5), [0.1, 0.1, 0.0, 0.0], mp.StateGoal([0.9, 0.9, 0.0, 0.0]), CC, ctx), dict(r=1.0)
    if name == "dubins":
        return mp.MPProblem(mp.DubinsQuasiMetricSpace(0.15), [0.1, 0.1, 0.0], mp.PointGoal([0.9, 0.9]), CC, ctx), dict(r=0.3, ensure_goal_ct=10)
    raise ValueError(name)


def solve(name, seed, ctx=None):
    """Cost of fmtstar!(P, 1000, ...) for the set-up with the samples drawn from numpy's PCG64(seed); None when the goal is not reached."""
    P, kw = problem(name, ctx)
    out = mp.fmtstar_(P, 1000, connections="R", rng=np.random.default_rng(seed), **kw)
    if not isinstance(out, tuple) or out[0] != "solved":
        return None
    return float(out[1])
