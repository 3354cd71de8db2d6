"""GPU tests of the host-side mirror of the reference's Julia interface (motionplanning.jl_amd/mirror.py): the
calls read like the reference's notebook (docs/MotionPlanning.ipynb cells 4-8) and results are checked against the
CPU oracle run on the samples the mirror drew."""
import json
import os

import numpy as np
import pytest

import motionplanning_jl_amd as mp

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def boxes2d():
    return [mp.BoxBounds(lo, hi) for lo, hi in json.load(open(os.path.join(G, "boxes_nd.json")))["BOXES2D"]]


def test_geometric_planning_like_the_notebook(orc):
    statespace = mp.UnitHypercube(2)
    init = [0.1, 0.1]
    goal = mp.BallGoal([0.9, 0.9], 0.05)
    collisionchecker = mp.PointRobotNDBoxes(boxes2d())
    P = mp.MPProblem(statespace, init, goal, collisionchecker)
    rng = np.random.default_rng(123)
    status, cost, elapsed = mp.fmtstar_(P, 2000, connections="R", rm=1.5, rng=rng)
    assert status == "solved" and len(P.V) == 2000
    assert np.array_equal(P.V[1], init) and np.array_equal(P.V.V[0], init)
    md = P.solution.metadata
    X, lohi = P.V.V, collisionchecker.lohi()
    assert orc.unpack(orc.points_free(X, lohi, statespace.lo, statespace.hi), len(X)).all()      # sampler keeps free states
    ref = orc.fmtstar(X, md["r"], orc.GOAL_BALL, goal.params(), lohi, statespace.lo, statespace.hi)
    assert ref["status"] == 1 and abs(ref["cost"] - cost) <= 1e-6 * cost
    assert np.array_equal(md["tree"] - 1, ref["A"]) and np.array_equal(md["path"] - 1, ref["path"])
    assert md["collision_checks"] == ref["collision_checks"] == collisionchecker.count
    assert cost >= np.linalg.norm(np.array([0.9, 0.9]) - init) - 0.05
    # cached and uncached neighbour queries agree with each other and with the oracle (nearneighbors.jl:120-136)
    inds, ds = mp.inball(P.V, 7, md["r"])
    oi, od = orc.inball(X, 6, md["r"])
    assert np.array_equal(inds - 1, oi) and np.allclose(ds, od, rtol=1e-6, atol=0)
    mp.build_cache_(P.V, md["r"])
    i2, d2 = mp.inball_(P.V, 7, md["r"])
    assert np.array_equal(i2, inds) and np.array_equal(d2, ds)
    W = np.ones(len(X), bool); W[inds[0] - 1] = False
    i3, _ = mp.inball_(P.V, 7, md["r"], W)
    assert np.array_equal(i3, inds[1:])


def test_planning_with_the_device_sampler_is_reproducible(orc):
    """fmtstar!(P, N) with the sampling loop on the device (seeded counter-based stream): same samples as the scalar
    loop, same plan every time."""
    SS = mp.UnitHypercube(2)
    out = []
    for _ in range(2):
        CC = mp.PointRobotNDBoxes(boxes2d())
        P = mp.MPProblem(SS, [0.1, 0.1], mp.BallGoal([0.9, 0.9], 0.05), CC)
        status, cost, _ = mp.fmtstar_(P, 3000, rm=1.5, seed=42, ensure_goal_ct=3)
        assert status == "solved" and len(P.V) == 3000
        out.append((cost, P.V.V.copy(), P.solution.metadata["path"].copy()))
    assert out[0][0] == out[1][0] and np.array_equal(out[0][1], out[1][1]) and np.array_equal(out[0][2], out[1][2])
    # MPProblem starts with V = [init] (statespaces.jl:163-170), so sample_free! draws N-1 new samples without an init slot
    rc, W, _ = orc.sample_free(42, 2999, 2, None, CC.lohi(), SS.lo, SS.hi, orc.GOAL_BALL, np.array([0.9, 0.9, 0.05]), goal_ct=3)
    assert rc == 0 and np.array_equal(out[0][1][0], [0.1, 0.1]) and np.array_equal(out[0][1][1:], W)


def test_scalar_validity_calls_and_count(orc):
    SS = mp.UnitHypercube(2)
    CC = mp.PointRobotNDBoxes(boxes2d())
    P = mp.MPProblem(SS, [0.1, 0.1], mp.PointGoal([0.9, 0.9]), CC)
    assert mp.is_free_state([0.1, 0.1], CC, SS, P.ctx) and not mp.is_free_state([0.45, 0.25], CC, SS, P.ctx)
    assert not mp.is_free_state([1.2, 0.5], CC, SS, P.ctx)                     # outside the state space
    CC.count = 0
    assert not mp.is_free_motion([0.30, 0.25], [0.60, 0.25], CC, SS, P.ctx)
    assert mp.is_free_motion([0.30, 0.10], [0.60, 0.10], CC, SS, P.ctx)
    assert CC.count == 2
    bigger = CC.inflate(0.2)
    assert len(bigger.boxes) == 5 and np.allclose(bigger.boxes[0].lo, CC.boxes[0].lo - 0.2)
    assert len(CC.addblocker([0.5, 0.5], 0.05).boxes) == 6


def test_infeasible_init_warns():
    SS = mp.UnitHypercube(2)
    P = mp.MPProblem(SS, [0.45, 0.25], mp.PointGoal([0.9, 0.9]), mp.PointRobotNDBoxes(boxes2d()))
    with pytest.warns(UserWarning, match="infeasible"):
        out = mp.fmtstar_(P, 100)
    assert out == np.inf and P.status == "failed"


def test_double_integrator_planning_like_the_notebook(orc):
    statespace = mp.DoubleIntegrator(2, vmax=0.5)
    init = [0.1, 0.1, 0.0, 0.0]
    goal = mp.StateGoal([0.9, 0.9, 0.0, 0.0])
    P = mp.MPProblem(statespace, init, goal, mp.PointRobotNDBoxes(boxes2d()))
    status, cost, _ = mp.fmtstar_(P, 1500, connections="R", r=1.0, rng=np.random.default_rng(7))
    X = P.V.V
    assert np.array_equal(X[-1], [0.9, 0.9, 0.0, 0.0])                         # goal sample in the tail (sampling.jl:38-42)
    lohi = P.CC.lohi()
    oc, orow, oval, _ = orc.di_pairwise(X, 1.0, 1.0)
    ref = orc.di_fmtstar(X, 1.0, 1.0, oc, orow, oval, orc.GOAL_POINT, X[-1], lohi, statespace.lo, statespace.hi)
    assert (status == "solved") == bool(ref["status"])
    assert np.array_equal(P.solution.metadata["tree"] - 1, ref["A"])
    assert P.solution.metadata["collision_checks"] == ref["collision_checks"]
    if ref["status"]:
        assert abs(cost - ref["cost"]) <= 1e-6 * ref["cost"]


def test_polygon_world_planning_like_the_notebook(orc):
    """The notebook's 2-D setup (PointRobot2D over ISRR_POLY-style obstacles, SAT2D.jl) through the mirror types."""
    import json, os
    fx = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "shapes_2d.json")))
    parts = [mp.Circle(s[1], s[2]) if s[0] == "circle" else mp.Polygon(s[1]) for s in fx["worlds"]["ISRR_POLY"]]
    CC = mp.PointRobot2D(mp.Compound2D(parts))
    SS = mp.UnitHypercube(2)
    P = mp.MPProblem(SS, [0.1, 0.1], mp.BallGoal([0.9, 0.9], 0.05), CC)
    status, cost, _ = mp.fmtstar_(P, 2500, rm=1.5, seed=9)
    assert status == "solved"
    md = P.solution.metadata
    X = P.V.V
    S = orc.Shapes2D([("circle", tuple(s[1]), s[2]) if s[0] == "circle" else ("polygon", [tuple(p) for p in s[1]])
                      for s in fx["worlds"]["ISRR_POLY"]])
    assert orc.unpack(orc.points_free_2d(X, S, SS.lo, SS.hi), len(X)).all()
    path = md["path"] - 1
    seg_free = orc.unpack(orc.motions_free_2d(X[path[:-1]], X[path[1:]], S, SS.lo, SS.hi), len(path) - 1)
    assert seg_free.all()                                         # the returned path is collision free
    assert abs(cost - np.sum(np.linalg.norm(X[path[1:]] - X[path[:-1]], axis=1))) <= 1e-9 * cost
    assert not mp.is_free_motion([0.3, 0.5], [0.6, 0.5], CC, SS, P.ctx)       # through the big hexagon
    assert mp.is_free_motion([0.05, 0.05], [0.95, 0.05], CC, SS, P.ctx)
    assert CC.count == md["collision_checks"] + 2
    blocked = CC.addblocker([0.5, 0.05], 0.04)
    assert not mp.is_free_motion([0.05, 0.05], [0.95, 0.05], blocked, SS, P.ctx)


def test_dubins_planning_through_the_mirror(orc):
    """DubinsQuasiMetricSpace (simplecars.jl:32-38) through the mirror types: SE2 samples, workspace goal lifted to a state,
    the plan checked against the oracle's Dubins graph / recursion on the samples the mirror drew."""
    rt = 0.08
    SS = mp.DubinsQuasiMetricSpace(rt)
    assert mp.dim(SS) == 3 and SS.workspace_dim == 2 and abs(mp.volume(SS) - 2 * np.pi) < 1e-12
    CC = mp.PointRobotNDBoxes(boxes2d())
    P = mp.MPProblem(SS, [0.1, 0.1, 0.5], mp.BallGoal([0.9, 0.9], 0.06), CC)
    status, cost, _ = mp.fmtstar_(P, 1800, rm=1.2, rng=np.random.default_rng(3), ensure_goal_ct=3)
    X = P.V.V
    assert X.shape == (1800, 3) and np.all((X[:, 2] >= 0) & (X[:, 2] <= 2 * np.pi))
    assert np.all(np.linalg.norm(X[-3:, :2] - [0.9, 0.9], axis=1) <= 0.06)          # goal samples in the tail
    md = P.solution.metadata
    lohi = CC.lohi()
    oc, orow, oval = orc.dubins_graph(X, rt, 1.0, md["r"])
    colptr, rowval, nzval = P.ctx.dubins_graph(rt, 1.0, md["r"])
    # (the oracle's sin / cos / atan2 / acos are the C library's, the device's are mp_math.h: costs agree to 1e-12, not bit for bit --
    # the recursion is pinned on the device's own edge costs so that no near-tie between two parents can break differently)
    assert np.array_equal(colptr - 1, oc) and np.array_equal(rowval - 1, orow) and np.allclose(nzval, oval, rtol=1e-12, atol=0)
    ref = orc.dubins_fmtstar(X, rt, 1.0, oc, orow, nzval, orc.GOAL_BALL, np.array([0.9, 0.9, 0.06]), lohi, SS.lo, SS.hi)
    assert (status == "solved") == bool(ref["status"])
    assert np.array_equal(md["tree"] - 1, ref["A"]) and md["collision_checks"] == ref["collision_checks"]
    if ref["status"]:
        assert abs(cost - ref["cost"]) <= 1e-9 * ref["cost"]
        path = md["path"] - 1
        for a, b in zip(path[:-1], path[1:]):                       # every edge of the plan is a free Dubins motion
            assert orc.dubins_is_free_motion(X[a], X[b], rt, 1.0, lohi, SS.lo, SS.hi)[0]


def test_reedsshepp_planning_through_the_mirror(orc):
    """ReedsSheppMetricSpace (simplecars.jl:29-34) through the mirror types; the plan is checked against the oracle's
    Reeds-Shepp graph / recursion on the samples the mirror drew, and never costs more than the Dubins plan's bound."""
    rt = 0.08
    SS = mp.ReedsSheppMetricSpace(rt)
    assert mp.dim(SS) == 3 and SS.workspace_dim == 2
    CC = mp.PointRobotNDBoxes(boxes2d())
    P = mp.MPProblem(SS, [0.1, 0.1, 0.5], mp.BallGoal([0.9, 0.9], 0.06), CC)
    status, cost, _ = mp.fmtstar_(P, 1500, rm=1.0, rng=np.random.default_rng(4), ensure_goal_ct=3)
    X = P.V.V
    md = P.solution.metadata
    lohi = CC.lohi()
    oc, orow, oval = orc.rs_graph(X, rt, 1.0, md["r"])
    colptr, rowval, nzval = P.ctx.reedsshepp_graph(rt, 1.0, md["r"])
    assert np.array_equal(colptr - 1, oc) and np.array_equal(rowval - 1, orow) and np.allclose(nzval, oval, rtol=1e-12, atol=0)
    # C|C|C words tie exactly in exact arithmetic (see test_reedsshepp_graph_sweep_and_plan): pin the recursion on the
    # device's own edge costs
    ref = orc.rs_fmtstar(X, rt, 1.0, oc, orow, nzval, orc.GOAL_BALL, np.array([0.9, 0.9, 0.06]), lohi, SS.lo, SS.hi)
    assert (status == "solved") == bool(ref["status"])
    assert np.array_equal(md["tree"] - 1, ref["A"]) and md["collision_checks"] == ref["collision_checks"]
    if ref["status"]:
        assert abs(cost - ref["cost"]) <= 1e-9 * ref["cost"]
        path = md["path"] - 1
        for a, b in zip(path[:-1], path[1:]):
            assert orc.car_is_free_motion(2, X[a], X[b], rt, 1.0, lohi, SS.lo, SS.hi)[0]


def test_closest_through_the_mirror(orc):
    """closest(p, CC, W) / closeR(p, CC, W, r2) with the mirror's collision checkers (boxesND.jl:33-34, robots2D.jl:25-26)."""
    ctx = mp.Context(0)
    CC = mp.PointRobotNDBoxes(boxes2d())
    W = np.array([[2.0, 0.3], [0.3, 0.5]])
    p = np.array([0.05, 0.5])
    d2, v = mp.closest(p, CC, W, ctx=ctx)
    od2, ov, ok, bad = orc.closest_boxes(p, CC.lohi(), W)
    assert bad == 0 and abs(d2 - od2[0]) <= 1e-12 and np.abs(v - ov[0]).max() <= 1e-12
    lst = mp.closeR(p, CC, W, 10.0)
    optr, oidx, odd, ovv = orc.closeR_boxes(p, CC.lohi(), W, 10.0)
    assert len(lst) == len(oidx) == len(CC.boxes) and np.allclose([c[0] for c in lst], odd, rtol=1e-12, atol=1e-15)
    C2 = mp.PointRobot2D(mp.Compound2D(mp.Circle((0.3, 0.4), 0.1), mp.Box2D((0.6, 0.8), (0.2, 0.5))))
    d2, v = mp.closest(np.array([0.3, 0.9]), C2, None, ctx=ctx)                      # Euclidean: towards the circle
    assert abs(d2 - 0.16) < 1e-14 and np.allclose(v, [0.3, 0.5], atol=1e-14)
    d2b, vb = mp.closest(np.array([[0.3, 0.9], [0.7, 0.0]]), C2, W)
    assert d2b.shape == (2,) and vb.shape == (2, 2)


def test_device_recursion_through_the_mirror(orc):
    """fmtstar_(..., band=b): the same call with the recursion on the device; a re-plan on the same samples (the reference's
    neighbour cache persisting in P.V across fmtstar! calls) with a band gives a valid plan whose cost is within a few percent."""
    P = mp.MPProblem(mp.UnitHypercube(2), [0.1, 0.1], mp.BallGoal([0.9, 0.9], 0.05), mp.PointRobotNDBoxes(boxes2d()))
    rng = np.random.default_rng(5)
    status, cost, _ = mp.fmtstar_(P, 3000, rm=1.5, rng=rng)
    seq_tree = P.solution.metadata["tree"].copy()
    status2, cost2, _ = mp.fmtstar_(P, 3000, rm=1.5, rng=rng, band=0.0)
    assert status == status2 == "solved" and len(P.V) == 3000
    assert cost2 >= cost * (1 - 1e-12) and cost2 <= cost * 1.05
    status3, cost3, _ = mp.fmtstar_(P, 3000, rm=1.5, rng=rng, band=0.5)
    assert status3 == "solved" and cost3 <= cost * 1.1
    X, lohi = P.V.V, P.CC.lohi()
    path = P.solution.metadata["path"]
    for a, b in zip(path[:-1], path[1:]):
        assert orc.is_free_motion(X[a - 1], X[b - 1], lohi, P.SS.lo, P.SS.hi)
    assert (P.solution.metadata["tree"] > 0).sum() > 0.5 * (seq_tree > 0).sum()
