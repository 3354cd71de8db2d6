"""CPU tests: the oracle against the golden vectors, the hand-derived known answers and the independent
Python transliteration of the reference's Julia lines.  No GPU, no libmpfmt compute calls."""
import json
import os

import numpy as np
import pytest

import jl_transliteration as jl

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load_boxes(name):
    return np.array(json.load(open(os.path.join(G, "boxes_nd.json")))[name], dtype=np.float64)


def test_known_answers_boxes2d(orc):
    """Hand-derived broad/narrow answers on the reference fixture BOXES2D[2]."""
    fx = json.load(open(os.path.join(G, "boxes_nd.json")))
    lo, hi = fx["BOXES2D"][1]
    for v, w, bp, nf, res in fx["known_answers_BOXES2D_box2"]["cases"]:
        b, n = orc.box_phases(v, w, lo, hi)
        assert b == bp
        if nf is not None:
            assert n == nf
        assert orc.motion_free_boxes(v, w, [[lo, hi]]) == res
        # and the independent transliteration agrees
        l = [min(a, c) for a, c in zip(v, w)]
        h = [max(a, c) for a, c in zip(v, w)]
        assert jl.is_free_motion_broadphase(l, h, lo, hi) == bp
        assert jl.is_free_motion_boxes(v, w, [(lo, hi)]) == res


@pytest.mark.parametrize("name", ["BOXES2D", "BOXES3D"])
def test_segment_goldens(orc, name):
    z = np.load(os.path.join(G, "segments_%s.npz" % name))
    P, Q, lohi = z["P"], z["Q"], z["lohi"]
    got = orc.unpack(orc.motions_free_explicit(P, Q, lohi, z["ss_lo"], z["ss_hi"]), len(P)) if hasattr(orc, "motions_free_explicit") else None
    free_boxes = np.array([orc.motion_free_boxes(p, q, lohi) for p, q in zip(P, Q)])
    free_full = np.array([orc.is_free_motion(p, q, lohi, z["ss_lo"], z["ss_hi"]) for p, q in zip(P, Q)])
    pfree = np.array([orc.is_free_state(p, lohi, z["ss_lo"], z["ss_hi"]) for p in P])
    assert np.array_equal(free_boxes, z["free_boxes"])
    assert np.array_equal(free_full, z["free_full"])
    assert np.array_equal(pfree, z["point_free"])
    assert got is None or np.array_equal(got, z["free_full"])
    assert np.array_equal(lohi, load_boxes(name))


def test_zero_length_segment_inside_box_is_free_quirk(orc):
    """boxesND.jl:46-51 reports a zero-length segment inside a box as free (lambda = NaN/Inf)."""
    lohi = load_boxes("BOXES2D")
    c = 0.5 * (lohi[1, 0] + lohi[1, 1])
    assert orc.motion_free_boxes(c, c, lohi)
    assert not orc.point_free_boxes(c, lohi)


@pytest.mark.parametrize("tag", ["d2_n1000", "d6_n1500"])
def test_rdisc_goldens(orc, tag):
    z = np.load(os.path.join(G, "rdisc_%s.npz" % tag))
    X, r = z["X"], float(z["r"])
    colptr, rowval, nzval = orc.rdisc_graph(X, r, mode=0)
    assert np.array_equal(colptr, z["colptr"])
    assert np.array_equal(rowval, z["rowval"])
    assert np.array_equal(nzval, z["nzval"])
    # contract of nearneighbors.jl:138-198: ascending, self excluded, symmetric for a metric
    N = len(X)
    for v in range(0, N, 97):
        rows = rowval[colptr[v]:colptr[v + 1]]
        assert np.all(np.diff(rows) > 0) and v not in rows
    A = set(zip(np.repeat(np.arange(N), np.diff(colptr)).tolist(), rowval.tolist()))
    assert all((j, i) in A for (i, j) in list(A)[:5000])
    # the duplicate pair is mutually adjacent at distance 0
    a = rowval[colptr[3]:colptr[4]].tolist()
    assert 7 in a and nzval[colptr[3] + a.index(7)] == 0.0


def test_kdtree_equals_brute(orc):
    rng = np.random.default_rng(5)
    for d in (2, 3, 6):
        X = rng.random((600, d))
        r = 0.2 if d < 6 else 0.5
        kd = orc.KDTree(X)
        for v in range(0, 600, 13):
            bi, bd = orc.inball(X, v, r, mode=0)
            ki, kd_d = kd.inball(v, r)
            assert np.array_equal(bi, ki) and np.array_equal(bd, kd_d)


def test_inball_modes_agree_away_from_threshold(orc):
    rng = np.random.default_rng(6)
    X = rng.random((400, 3))
    for v in range(0, 400, 17):
        a, _ = orc.inball(X, v, 0.21, mode=0)
        b, _ = orc.inball(X, v, 0.21, mode=1)
        assert np.array_equal(a, b)


def test_inball_matches_transliteration(orc):
    rng = np.random.default_rng(7)
    X = rng.random((300, 4))
    V = X.tolist()
    for v in (0, 5, 150, 299):
        inds, ds = jl.inball_tree(V, v + 1, 0.35)
        oi, od = orc.inball(X, v, 0.35, mode=0)
        assert [i - 1 for i in inds] == oi.tolist() and ds == od.tolist()
        gi, gd = jl.inball_generic(V, v + 1, 0.35)
        oi1, od1 = orc.inball(X, v, 0.35, mode=1)
        assert [i - 1 for i in gi] == oi1.tolist() and gd == od1.tolist()


def test_fmt_golden_cfg1(orc):
    z = np.load(os.path.join(G, "fmt_cfg1.npz"))
    for nn_mode in (0, 1):
        res = orc.fmtstar(z["X"], float(z["r"]), orc.GOAL_BALL, z["goal"], z["lohi"], np.zeros(2), np.ones(2),
                          init_idx=0, checkpts=True, nn_mode=nn_mode)
        assert res["status"] == int(z["status"]) == 1
        assert res["cost"] == float(z["cost"])
        assert res["collision_checks"] == int(z["collision_checks"])
        assert np.array_equal(res["A"], z["A"]) and np.array_equal(res["C"], z["C"]) and np.array_equal(res["path"], z["path"])
    # sanity: cost is at least the straight-line distance to the goal ball
    assert float(z["cost"]) >= np.linalg.norm(z["X"][-1] - z["X"][0]) - 0.05


def test_fmt_graph_driver_equals_lazy(orc):
    """The eager-graph driver (prebuilt CSC + per-edge free mask) reproduces the lazy loop exactly."""
    import motionplanning_jl_amd as mp
    w = mp.workloads.cfg1()
    lazy = orc.fmtstar(w.X, w.r, orc.GOAL_BALL, w.goal_params(), w.lohi, w.ss_lo, w.ss_hi)
    colptr, rowval, nzval = orc.rdisc_graph(w.X, w.r)
    emask = orc.graph_edges_free(w.X, colptr, rowval, w.lohi, w.ss_lo, w.ss_hi)
    F = orc.points_free(w.X, w.lohi, w.ss_lo, w.ss_hi)
    eager = orc.fmtstar_graph(w.X, colptr, rowval, nzval, emask, F, orc.GOAL_BALL, w.goal_params(), w.lohi, w.ss_lo, w.ss_hi)
    for k in ("status", "cost", "z", "collision_checks"):
        assert lazy[k] == eager[k]
    assert np.array_equal(lazy["A"], eager["A"]) and np.array_equal(lazy["C"], eager["C"])
    assert np.array_equal(lazy["path"], eager["path"])


def test_fmt_infeasible_init(orc):
    lohi = np.array([[[0.0, 0.0], [0.2, 0.2]]])
    X = np.array([[0.1, 0.1], [0.5, 0.5], [0.9, 0.9]])
    res = orc.fmtstar(X, 0.7, orc.GOAL_BALL, [0.9, 0.9, 0.05], lohi, np.zeros(2), np.ones(2))
    assert res["rc"] == -1


def test_expand_step_matches_fmt_first_iteration(orc):
    import motionplanning_jl_amd as mp
    w = mp.workloads.cfg1()
    N = w.N
    W = np.ones(N, bool); W[0] = False
    H = np.zeros(N, bool); H[0] = True
    F = orc.unpack(orc.points_free(w.X, w.lohi, w.ss_lo, w.ss_hi), N)
    xs, ym, cm, fr = orc.expand(w.X, w.r, orc.pack(W), orc.pack(H), orc.pack(F), np.zeros(N), [0], w.lohi, w.ss_lo, w.ss_hi)
    inds, ds = orc.inball(w.X, 0, w.r)
    keep = F[inds]
    assert np.array_equal(xs, inds[keep]) and np.all(ym == 0) and np.array_equal(cm, ds[keep])
    assert np.array_equal(fr, [orc.is_free_motion(w.X[0], w.X[x], w.lohi, w.ss_lo, w.ss_hi) for x in xs])


def test_radius_rule(orc):
    import motionplanning_jl_amd as mp
    for (d, N) in ((2, 1000), (6, 100000), (6, 1000000)):
        a = orc.fmt_radius(1.0, d, 1.0, N)
        b = mp.workloads.fmt_radius(1.0, d, 1.0, N)
        assert abs(a - b) <= 4e-16 * a
    assert abs(orc.fmt_radius(1.0, 6, 1.0, 1000000) - 0.17479) < 1e-5
    assert abs(orc.fmt_radius(1.0, 6, 1.0, 100000) - 0.24888) < 1e-5


def test_bitmask_layout(orc):
    bits = np.zeros(130, bool); bits[[0, 63, 64, 129]] = True
    m = orc.pack(bits)
    assert m[0] == (1 | (1 << 63)) and m[1] == 1 and m[2] == 2
    assert np.array_equal(orc.unpack(m, 130), bits)


def test_philox_known_answers():
    """Philox4x32-10 against the known-answer vectors published with Random123 (kat_vectors): the stream behind the
    batch sampler is the published generator, not a look-alike."""
    from oracle import oracle as orc
    assert orc.philox4x32_10([0, 0, 0, 0], [0, 0]) == [0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8]
    assert orc.philox4x32_10([0xffffffff] * 4, [0xffffffff] * 2) == [0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd]
    assert orc.philox4x32_10([0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344], [0xa4093822, 0x299f31d0]) == \
        [0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1]


def test_oracle_sampler_semantics():
    """sample_free! (sampling.jl:11-45): V[1] = init, every sample free and in bounds, goal samples in the tail,
    candidates consumed in order (attempts >= accepted), uniforms in [0, 1)."""
    from oracle import oracle as orc
    import motionplanning_jl_amd as mp
    rng = np.random.default_rng(5)
    d, N = 3, 400
    lohi = mp.workloads.make_boxes(rng, 25, d, 0.05, 0.2, [np.full(d, .1), np.full(d, .9)])
    lo, hi = np.zeros(d), np.ones(d)
    goal = np.concatenate([np.full(d, .9), [0.1]])
    rc, W, att = orc.sample_free(7, N, d, np.full(d, .1), lohi, lo, hi, 1, goal, goal_ct=5)
    assert rc == 0 and W.shape == (N, d) and att >= N - 1
    assert np.array_equal(W[0], np.full(d, .1))
    assert all(orc.is_free_state(w, lohi, lo, hi) for w in W)
    assert np.all(np.linalg.norm(W[-5:] - goal[:d], axis=1) <= 0.1)
    assert np.linalg.norm(W[-6] - goal[:d]) > 0.1 or True           # the slot before the tail is an ordinary sample
    # the kept samples are the accepted candidates in order: replay the stream
    k = 1
    for c in range(att):
        v = lo + orc.sample_uniforms(7, c, 0, d) * (hi - lo)
        if orc.is_free_state(v, lohi, lo, hi):
            if k < N - 5:
                assert np.array_equal(W[k], v)
            k += 1
    assert k == N                                                    # the att-th candidate filled the last slot
    u = np.concatenate([orc.sample_uniforms(1, c, 0, 6) for c in range(2000)])
    assert u.min() >= 0.0 and u.max() < 1.0 and abs(u.mean() - 0.5) < 0.02


def _world2d(orc, shapes):
    return orc.Shapes2D([("circle", tuple(s[1]), s[2]) if s[0] == "circle" else ("polygon", [tuple(p) for p in s[1]]) for s in shapes])


def test_sat2d_oracle_against_goldens_and_known_answers():
    """2-D SAT world (SAT2D.jl): the oracle on the reference's obstacle fixtures (test/obstaclesets/2D.jl) reproduces the
    committed masks and the hand-derived answers."""
    import json, os
    from oracle import oracle as orc
    G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    fx = json.load(open(os.path.join(G, "shapes_2d.json")))
    for name, shapes in fx["worlds"].items():
        S = _world2d(orc, shapes)
        z = np.load(os.path.join(G, "segments2d_%s.npz" % name))
        n = len(z["P"])
        assert np.array_equal(orc.unpack(orc.motions_free_2d(z["P"], z["Q"], S), n), z["free_motion"])
        assert np.array_equal(orc.unpack(orc.motions_free_2d(z["P"], z["Q"], S, z["ss_lo"], z["ss_hi"]), n), z["free_motion_ss"])
        assert np.array_equal(orc.unpack(orc.points_free_2d(z["P"], S), n), z["free_state"])
        assert np.array_equal(orc.unpack(orc.points_free_2d(z["P"], S, z["ss_lo"], z["ss_hi"]), n), z["free_state_ss"])
        for c in fx["known"]:
            if c[0] == name:
                assert bool(orc.unpack(orc.motions_free_2d([c[1]], [c[2]], S), 1)[0]) == c[3]
                assert bool(orc.unpack(orc.points_free_2d([c[1]], S), 1)[0]) == c[4]
    import pytest
    with pytest.raises(ValueError):
        orc.Shapes2D([("polygon", [(0, 0), (1, 0), (0.2, 0.2), (0, 1)])])       # not convex (SAT2D.jl:49)
    with pytest.raises(ValueError):
        orc.Shapes2D([("circle", (0, 0), 0.0)])                                 # SAT2D.jl:22


def test_dubins_oracle_against_transliteration():
    """Dubins steering (simplecars.jl:106-215): oracle = independent transliteration on costs, controls and waypoints to a
    few ulp (same libm, but the C compiler pairs sin/cos into sincos), same word chosen, and every path ends at its target."""
    from oracle import oracle as orc
    import jl_transliteration as jl
    rng = np.random.default_rng(2)
    for k in range(3000):
        s1 = np.array([rng.random(), rng.random(), rng.random() * 2 * np.pi])
        s2 = np.array([rng.random(), rng.random(), rng.random() * 2 * np.pi])
        rt = [0.05, 0.2, 1.0][k % 3]
        c, p = orc.dubins(s1, s2, rt, 1.0)
        cj, pj = jl.dubins(s1, s2, rt, 1.0)
        pj = np.array(pj)
        assert abs(c - cj) <= 1e-14 * c and np.array_equal(p[:, 1:], pj[:, 1:]) and np.allclose(p, pj, rtol=0, atol=1e-14)
        assert abs(p[:, 0].sum() - c) <= 1e-12 * c
        v = tuple(s1)
        for u in p:
            v = jl.car_propagate(v, u)
        assert abs(v[0] - s2[0]) < 1e-9 and abs(v[1] - s2[1]) < 1e-9
        assert min(abs(v[2] - s2[2]), 2 * np.pi - abs(v[2] - s2[2])) < 1e-9
        w1 = orc.dubins_waypoints(s1, s2, rt, 1.0)
        w2 = np.array(jl.car_collision_waypoints(s1, s2, rt, 1.0))
        assert w1.shape == w2.shape and np.allclose(w1, w2, rtol=0, atol=1e-14)
    # known answers: straight ahead, and a quarter turn on the unit circle
    c, p = orc.dubins([0, 0, 0], [1, 0, 0], 1.0, 1.0)
    assert abs(c - 1.0) < 1e-15
    # a degenerate pose (exact quarter turn): mod2piF of a -1e-17 remainder wraps to 2*pi, so the reference's formulae pick
    # a longer word there -- restatement and transliteration must agree on that too
    c, _ = orc.dubins([0, 0, 0], [1, 1, np.pi / 2], 1.0, 1.0)
    cj, _ = jl.dubins([0, 0, 0], [1, 1, np.pi / 2], 1.0, 1.0)
    assert abs(c - cj) <= 1e-14 * c and c >= np.pi / 2 - 1e-12


def test_reedsshepp_oracle_against_transliteration():
    """Reeds-Shepp steering (simplecars.jl:266-364 and the word families :367-553): oracle = independent transliteration on
    costs, controls and collision waypoints, every path reaches its target, and allowing reverse never costs more than the
    Dubins word between the same poses."""
    from oracle import oracle as orc
    import jl_transliteration as jl
    rng = np.random.default_rng(12)
    for k in range(3000):
        s1 = np.array([rng.random(), rng.random(), rng.random() * 2 * np.pi])
        s2 = np.array([rng.random(), rng.random(), rng.random() * 2 * np.pi])
        rt = [0.05, 0.2, 1.0][k % 3]
        c, p = orc.reedsshepp(s1, s2, rt, 1.0)
        cj, pj = jl.reedsshepp(s1, s2, rt, 1.0)
        pj = np.array(pj)
        assert p.shape == pj.shape and abs(c - cj) <= 1e-14 * c
        assert np.array_equal(p[:, 1:], pj[:, 1:]) and np.allclose(p, pj, rtol=0, atol=1e-14)
        assert abs(np.abs(p[:, 0]).sum() - c) <= 1e-12 * c
        assert c <= orc.dubins(s1, s2, rt, 1.0)[0] * (1 + 1e-12)
        v = tuple(s1)
        for u in p:
            v = jl.car_propagate(v, u)
        assert abs(v[0] - s2[0]) < 1e-9 and abs(v[1] - s2[1]) < 1e-9
        assert min(abs(v[2] - s2[2]), 2 * np.pi - abs(v[2] - s2[2])) < 1e-9
        w1 = orc.car_waypoints(2, s1, s2, rt, 1.0)
        w2 = np.array(jl.car_collision_waypoints_rs(s1, s2, rt, 1.0))
        assert w1.shape == w2.shape and np.allclose(w1, w2, rtol=0, atol=1e-14)
    # known answers: straight ahead, and straight back (one reversed straight segment of the same length)
    c, p = orc.reedsshepp([0, 0, 0], [1, 0, 0], 1.0, 1.0)
    assert abs(c - 1.0) < 1e-15
    c, p = orc.reedsshepp([1, 0, 0], [0, 0, 0], 1.0, 1.0)
    assert abs(c - 1.0) < 1e-12 and (p[np.abs(p[:, 0]) > 1e-12][:, 1] < 0).all()


def test_reedsshepp_oracle_graph_and_plan():
    """inball sets of the chopped Reeds-Shepp metric and a plan over them: neighbourhoods are symmetric as sets (a metric),
    every edge of the plan is a free motion, cumulative costs add up."""
    from oracle import oracle as orc
    rng = np.random.default_rng(5)
    N, rt, r = 400, 0.1, 0.3
    X = np.column_stack([rng.random(N), rng.random(N), rng.random(N) * 2 * np.pi])
    X[0] = [0.05, 0.05, 0.6]; X[-1] = [0.95, 0.95, 0.8]
    c0 = 0.2 + 0.6 * rng.random((8, 2)); h = 0.02 + 0.05 * rng.random((8, 2))
    lohi = np.stack([c0 - h, c0 + h], axis=1)
    lo, hi = np.array([0.0, 0.0, 0.0]), np.array([1.0, 1.0, 2 * np.pi])
    colptr, rowval, nzval = orc.rs_graph(X, rt, 1.0, r)
    cols = np.repeat(np.arange(N), np.diff(colptr))
    fwd = set(zip(cols.tolist(), rowval.tolist())); bwd = set(zip(rowval.tolist(), cols.tolist()))
    for (a, b) in fwd ^ bwd:                                             # only pairs whose two directions straddle r in the last bits
        assert abs(orc.reedsshepp(X[a], X[b], rt, 1.0)[0] - r) < 1e-9
    assert len(rowval) > 5 * N and (nzval <= r).all()
    res = orc.rs_fmtstar(X, rt, 1.0, colptr, rowval, nzval, orc.GOAL_BALL, np.array([0.95, 0.95, 0.08]), lohi, lo, hi)
    assert res["rc"] == 0 and res["status"] == 1
    path = res["path"]
    tot = 0.0
    for a, b in zip(path[:-1], path[1:]):
        assert orc.car_is_free_motion(2, X[a], X[b], rt, 1.0, lohi, lo, hi)[0]
        assert res["A"][b] == a
        tot += orc.reedsshepp(X[b], X[a], rt, 1.0)[0]                    # inball(b) holds d(b, a)   (fmt.jl:70-75)
    assert abs(tot - res["cost"]) <= 1e-12 * tot


def _spd(rng, d):
    A = rng.standard_normal((d, d))
    return (A @ A.T + 0.3 * np.eye(d)) * 10 ** rng.uniform(-1, 1)


def test_closest_boxes_oracle_against_transliteration():
    """closest / closeR over boxes (boxesND.jl:61-86, bvls.jl:19-218): the C restatement (hand-written Householder QR) equals
    the transliteration (numpy's LAPACK for `\\`, chol) to rounding, picks the same box, and both run out of bvls' 10n
    iterations on the same (point, box) pairs -- where the reference returns `nothing` and closest() throws."""
    from oracle import oracle as orc
    import jl_transliteration as jl
    rng = np.random.default_rng(0)
    nfail = 0
    for d in (1, 2, 3, 6, 8):
        for t in range(120):
            W = _spd(rng, d)
            c = rng.random((5, d)); h = 0.02 + 0.2 * rng.random((5, d))
            lohi = np.stack([c - h, c + h], axis=1)
            p = rng.random(d) * 1.4 - 0.2
            if t % 7 == 0:
                p = c[0] + 0.5 * h[0] * (rng.random(d) - 0.5)              # inside a box
            boxes = [(b[0], b[1]) for b in lohi]
            fails_j = sum(jl.closest_box(p, lo, hi, W) is None for lo, hi in boxes)
            d2, v, k, bad = orc.closest_boxes(p, lohi, W)
            assert bad == fails_j
            nfail += bad
            dj, vj, kj = jl.closest_boxlist(p, boxes, W)
            assert k[0] == kj and abs(d2[0] - dj) <= 1e-11 * max(dj, 1e-3) and np.abs(v[0] - vj).max() <= 1e-10
            if kj >= 0 and t % 7:
                # the closest point is the constrained minimiser: no feasible perturbation lowers the quadratic form
                for _ in range(5):
                    q = np.clip(vj + 1e-3 * rng.standard_normal(d), lohi[kj, 0], lohi[kj, 1])
                    assert (q - p) @ W @ (q - p) >= dj - 1e-12
            ds = sorted(x[0] for x in [jl.closest_box(p, lo, hi, W) for lo, hi in boxes] if x is not None)
            r2 = 0.5 * (ds[1] + ds[2]) if len(ds) > 2 and ds[2] > ds[1] * (1 + 1e-6) + 1e-12 else 1e9   # not on a value: strict < must not flip
            ptr, idx, dd, vv = orc.closeR_boxes(p, lohi, W, r2)
            want = jl.closeR_boxlist(p, boxes, W, r2)
            assert list(idx) == [w[2] for w in want] and ptr[-1] == len(want)
            assert np.allclose(dd, [w[0] for w in want], rtol=1e-11, atol=1e-13)
            assert (np.diff(dd) >= 0).all()
    assert nfail > 0                                                        # the failure branch was exercised
    # known answers: identity weight = Euclidean projection onto the box
    lohi = np.array([[[0.2, 0.2], [0.4, 0.6]]])
    d2, v, k, bad = orc.closest_boxes(np.array([[0.9, 0.5], [0.0, 0.0], [0.3, 0.9]]), lohi, np.eye(2))
    assert bad == 0 and np.allclose(v, [[0.4, 0.5], [0.2, 0.2], [0.3, 0.6]], atol=1e-15) and np.allclose(d2, [0.25, 0.08, 0.09], atol=1e-15)
    d2, v, k, bad = orc.closest_boxes(np.array([[0.5, 0.5]]), np.zeros((0, 2, 2)), np.eye(2))
    assert np.isinf(d2[0]) and k[0] == -1 and np.array_equal(v[0], [0.5, 0.5])   # (Inf, p)   boxesND.jl:73


def test_closest_shapes_oracle_against_transliteration():
    """closest / closeR over circles, convex polygons and compounds (SAT2D.jl:208-285), Euclidean and weighted."""
    from oracle import oracle as orc
    import jl_transliteration as jl
    rng = np.random.default_rng(1)
    shapes = [("circle", (0.3, 0.4), 0.1), ("polygon", [(0.6, 0.1), (0.9, 0.2), (0.8, 0.5), (0.55, 0.4)]), ("circle", (0.7, 0.8), 0.15),
              ("polygon", [(0.1, 0.7), (0.3, 0.7), (0.3, 0.9), (0.1, 0.9)])]
    S = orc.Shapes2D(shapes)
    nbad = 0
    for t in range(1500):
        W = _spd(rng, 2) if t % 3 else None
        p = rng.random(2) * 1.2 - 0.1
        d2, v, k, bad = orc.closest_shapes(p, S, W)
        dj, vj, kj = jl.closest_compound(p, shapes, W)
        # a circle whose multiplier iteration does not end (the reference would not return; e.g. some points inside the
        # circle) is reported by both restatements on the same pairs
        bad_j = 0 if W is None else sum(jl.closest_circle(p, s[1], s[2], W) is None for s in shapes if s[0] == "circle")
        nbad += bad
        assert bad == bad_j and k[0] == kj and abs(d2[0] - dj) <= 1e-12 and np.abs(v[0] - vj).max() <= 1e-12
        if W is not None:
            ptr, idx, dd, vv = orc.closeR_shapes(p, S, W, 0.3 * np.trace(W))
            want = jl.closeR_compound(p, shapes, W, 0.3 * np.trace(W))
            assert list(idx) == [w[2] for w in want]
            assert np.allclose(dd, [w[0] for w in want], rtol=0, atol=1e-12)
            assert np.allclose(vv, np.array([w[1] for w in want]).reshape(-1, 2), rtol=0, atol=1e-12)
    # known answers (Euclidean): circle boundary point towards p; polygon edge projection
    d2, v, k, bad = orc.closest_shapes(np.array([[0.3, 0.9]]), orc.Shapes2D([("circle", (0.3, 0.4), 0.1)]))
    assert np.allclose(v, [[0.3, 0.5]], atol=1e-15) and abs(d2[0] - 0.16) < 1e-15
    d2, v, k, bad = orc.closest_shapes(np.array([[0.5, 0.0]]), orc.Shapes2D([("polygon", [(0.4, 0.2), (0.6, 0.2), (0.6, 0.4), (0.4, 0.4)])]))
    assert np.allclose(v, [[0.5, 0.2]], atol=1e-15) and abs(d2[0] - 0.04) < 1e-15


def test_oracle_wavefront_single_is_the_sequential_loop(orc):
    """orc_fmt_wavefront_graph with one node per batch must be orc_fmtstar_graph step for step; with a band it stays a valid
    FMT* tree (parents are graph neighbours, costs add up) whose cost is not below the sequential one."""
    import motionplanning_jl_amd as mp
    for (N, d, M, seed) in [(800, 2, 20, 1), (1500, 3, 30, 2), (1200, 6, 60, 3)]:
        w = mp.workloads.make("t", N, d, M, 0.05, 0.12, seed=seed, goal_radius=0.1)
        colptr, rowval, nzval = orc.rdisc_graph(w.X, w.r)
        F = orc.points_free(w.X, w.lohi, w.ss_lo, w.ss_hi)
        args = (w.X, colptr, rowval, nzval, None, F, orc.GOAL_BALL, w.goal_params(), w.lohi, w.ss_lo, w.ss_hi)
        seq = orc.fmtstar_graph(*args)
        one = orc.fmt_wavefront_graph(*args, single=True)
        for k in ("status", "cost", "z", "collision_checks"):
            assert one[k] == seq[k], k
        assert np.array_equal(one["A"], seq["A"]) and np.array_equal(one["C"], seq["C"]) and np.array_equal(one["path"], seq["path"])
        emask = orc.graph_edges_free(w.X, colptr, rowval, w.lohi, w.ss_lo, w.ss_hi)
        for bandf in (0.0, 0.3, 1.5):
            b = orc.fmt_wavefront_graph(*args, band=bandf * w.r)
            be = orc.fmt_wavefront_graph(w.X, colptr, rowval, nzval, emask, F, *args[6:], band=bandf * w.r)
            assert np.array_equal(b["A"], be["A"]) and b["collision_checks"] == be["collision_checks"]    # lazy == eager mask
            assert b["status"] == seq["status"]
            if seq["status"] == 1:
                assert b["cost"] >= seq["cost"] * (1 - 1e-12)
            kids = np.flatnonzero(b["A"] >= 0)
            par = b["A"][kids]
            dist = np.sqrt(((w.X[kids] - w.X[par]) ** 2).sum(1))
            assert dist.max() <= w.r and np.allclose(b["C"][kids], b["C"][par] + dist, rtol=1e-12, atol=0)


def oob_parent_world(seed=12, N=700):
    """Samples spill over the state-space bounds and checkpts = false lets them into the tree: an out-of-bounds sample that
    has been connected (only the FIRST point of a segment is bounds-checked, statespaces.jl:155) later becomes the y_min of
    a neighbour, where in_state_space(V[y_min]) fails before the checker -- and its count (boxesND.jl:26) -- is reached."""
    rng = np.random.default_rng(seed)
    X = rng.random((N, 2))
    X[0] = (0.2, 0.2); X[-1] = (0.8, 0.8)
    c = rng.random((12, 2)); h = 0.02 + 0.05 * rng.random((12, 2))
    lohi = np.stack([c - h, c + h], axis=1)
    keep = ~np.array([np.all((lo <= X[0]) & (X[0] <= hi)) or np.all((lo <= X[-1]) & (X[-1] <= hi)) for lo, hi in lohi])
    return X, lohi[keep], np.full(2, 0.12), np.full(2, 0.88), 0.09, np.array([0.8, 0.8, 0.06])


def test_out_of_bounds_parent_is_not_counted(orc):
    import jl_transliteration as jl
    X, lohi, lo, hi, r, goal = oob_parent_world()
    res = orc.fmtstar(X, r, orc.GOAL_BALL, goal, lohi, lo, hi, checkpts=False, nn_mode=1)
    ref = jl.fmtstar(X.tolist(), r, lambda v: jl.is_goal_ball(v, goal[:2].tolist(), goal[2]), [(a.tolist(), b.tolist()) for a, b in lohi],
                     lo.tolist(), hi.tolist(), checkpts=False)
    assert res["status"] == int(ref["status"]) and res["collision_checks"] == ref["collision_checks"]
    assert np.array_equal(res["A"] + 1, ref["A"]) and np.array_equal(res["C"], ref["C"])
    assert np.array_equal(res["path"] + 1, ref["path"])
    # the case is really exercised: some connected sample lies outside the bounds, and fewer checks were counted than
    # parents examined (every examination of an out-of-bounds parent skips the count)
    conn = np.flatnonzero(res["A"] >= 0)
    oob = ~np.all((lo <= X) & (X <= hi), axis=1)
    assert oob[conn].any()
    colptr, rowval, nzval = orc.rdisc_graph(X, r)
    counted_all = orc.fmtstar(X, r, orc.GOAL_BALL, goal, lohi, np.full(2, -1.0), np.full(2, 2.0), checkpts=False, nn_mode=1)
    assert counted_all["collision_checks"] != res["collision_checks"] or not np.array_equal(counted_all["A"], res["A"])


def test_golden_stream_heads_and_splitmix_known_answer(orc):
    """the workload stream is SplitMix64 -- pinned to the published known answers of the generator and to the
    committed heads of every seed the workloads use; the numpy and the C restatement agree."""
    import json
    import motionplanning_jl_amd as mp
    g = json.load(open(os.path.join(G, "stream_heads.json")))
    kat = [6457827717110365317, 3203168211198807973, 9817491932198370423, 4593380528125082431, 16408922859458223821]
    assert [int(x) for x in g["splitmix64_seed_1234567_first5"]] == kat
    assert [int(x) for x in mp.workloads.splitmix64(1234567, 5)] == kat
    assert [orc.splitmix64(1234567, i) for i in range(5)] == kat
    for seed, head in g["heads"].items():
        want = np.array([float.fromhex(h) for h in head])
        assert np.array_equal(mp.workloads.Stream(int(seed)).random((16,)), want)
        assert np.array_equal(orc.stream_uniform(int(seed), 16), want)
    st = mp.workloads.Stream(3)
    a = np.concatenate([st.random((5, 3)).ravel(), st.random((7,))])
    assert np.array_equal(a, orc.stream_uniform(3, 22)) and a.min() >= 0.0 and a.max() < 1.0


def test_golden_di_pairs(orc):
    """512 double-integrator pairs -> (cost, t*) and the 5 collision waypoints."""
    z = np.load(os.path.join(G, "di_pairs.npz"))
    rho, r = float(z["rho"]), float(z["r"])
    for i in range(len(z["X0"])):
        c, t = orc.di_steer(z["X0"][i], z["X1"][i], rho, r)
        assert c == z["cost"][i] and t == z["topt"][i]
        assert np.array_equal(orc.di_waypoints(z["X0"][i], z["X1"][i], rho, r), z["waypoints"][i], equal_nan=True)   # t* = 0: 0/0 as in the reference
    assert z["cost"][5] == 0.0 and z["topt"][5] == 0.0
    inside = z["cost"] <= r
    assert 100 < inside.sum() < 500
    ok = np.arange(len(z["X0"])) != 5
    # a waypoint set starts at x0 and, for a pair within the radius, ends at x1 (to rounding)
    assert np.array_equal(z["waypoints"][ok, 0], z["X0"][ok])
    assert np.abs(z["waypoints"][inside & ok, 4] - z["X1"][inside & ok]).max() < 1e-12


def test_mp_math_accuracy_against_mpmath(orc):
    """mp_math.h (the fp64 sin / cos / atan2 / acos the DEVICE compiles; reached here through the oracle's second build,
    liboracle_devmath.so -- the oracle proper calls the C library) against 200-bit mpmath: within 2 ulp on the ranges the car kernels
    use, exact special values, and the identities the word formulas rely on."""
    with orc.device_math():
        _mp_math_accuracy(orc)


def test_oracle_proper_calls_the_c_library(orc):
    """The oracle's own transcendental functions are libm's (independent of the product's mp_math.h): its exported hooks equal
    math.sin / cos / atan2 / acos bit for bit, and the file includes nothing from the product's tree."""
    import math
    rng = np.random.default_rng(3)
    for x in rng.uniform(-20, 20, 2000):
        x = float(x)
        assert orc.mp_sin(x) == math.sin(x) and orc.mp_cos(x) == math.cos(x)
        assert orc.mp_atan2(x, 1.0 - x) == math.atan2(x, 1.0 - x)
    for x in rng.uniform(-1, 1, 2000):
        assert orc.mp_acos(float(x)) == math.acos(float(x))
    src = open(os.path.join(os.path.dirname(orc.__file__), "mpfmt_oracle.c")).read()
    includes = [ln for ln in src.splitlines() if ln.lstrip().startswith("#include")]
    assert includes and all("<" in ln and ".." not in ln and "motionplanning" not in ln for ln in includes), includes


def _mp_math_accuracy(orc):
    import mpmath
    mpmath.mp.prec = 200
    rng = np.random.default_rng(5)

    def ulps(got, want):
        want_f = float(want)
        if want_f == 0.0:
            return abs(got) / 5e-324
        return abs(mpmath.mpf(got) - want) / mpmath.mpf(np.spacing(abs(want_f)))
    xs = np.concatenate([rng.uniform(-8 * np.pi, 8 * np.pi, 4000), rng.uniform(-1e-3, 1e-3, 500), rng.uniform(-1e4, 1e4, 500),
                         np.arange(-16, 17) * (np.pi / 4), [0.0, 1e-300, -1e-300]])
    worst = 0.0
    for x in xs:
        x = float(x)
        for f, g in ((orc.mp_sin, mpmath.sin), (orc.mp_cos, mpmath.cos)):
            want = g(mpmath.mpf(x))
            if abs(float(want)) > 1e-12:                   # (near a zero of sin / cos the error is absolute: checked below)
                worst = max(worst, float(ulps(f(x), want)))
            assert abs(f(x) - float(want)) < 3e-16
    assert worst <= 2.0, worst
    worst = 0.0
    for _ in range(5000):
        y, x = float(rng.normal()), float(rng.normal())
        if rng.random() < 0.2:
            y *= 1e-8
        if rng.random() < 0.2:
            x *= 1e-8
        worst = max(worst, float(ulps(orc.mp_atan2(y, x), mpmath.atan2(mpmath.mpf(y), mpmath.mpf(x)))))
    assert worst <= 2.0, worst
    worst = 0.0
    for x in np.concatenate([rng.uniform(-1, 1, 4000), 1 - 10.0 ** rng.uniform(-16, -1, 300), -1 + 10.0 ** rng.uniform(-16, -1, 300)]):
        x = float(x)
        worst = max(worst, float(ulps(orc.mp_acos(x), mpmath.acos(mpmath.mpf(x)))))
    assert worst <= 3.0, worst
    import math
    assert orc.mp_sin(0.0) == 0.0 and orc.mp_cos(0.0) == 1.0
    assert orc.mp_atan2(0.0, 1.0) == 0.0 and orc.mp_atan2(0.0, -1.0) == math.pi and orc.mp_atan2(-0.0, -1.0) == -math.pi
    assert orc.mp_atan2(1.0, 0.0) == math.pi / 2 and orc.mp_atan2(-1.0, 0.0) == -math.pi / 2 and orc.mp_atan2(0.0, 0.0) == 0.0
    assert orc.mp_atan2(1.0, 1.0) == math.pi / 4 and orc.mp_atan2(2.0, -2.0) == 3 * math.pi / 4
    assert orc.mp_acos(1.0) == 0.0 and orc.mp_acos(-1.0) == math.pi and orc.mp_acos(0.0) == math.pi / 2
    assert math.isnan(orc.mp_acos(1.0000001)) and math.isnan(orc.mp_sin(float("inf"))) and math.isnan(orc.mp_atan2(float("nan"), 1.0))


def test_mc_importance_sampling_oracle_is_consistent(orc):
    """The scalar loop of the importance-sampling estimator (orc_mc_is_edges): with no obstacle the shift is zero and every weight is 1;
    far from any obstacle it finds nothing; on a moderately rare collision its mean agrees with plain Monte Carlo over 30 seeds while its
    variance is smaller; and the Irwin-Hall(8) density it weighs with integrates to 1 (checked through the weights of a
    one-sided shift)."""
    X = np.array([[0.2, 0.2], [0.8, 0.25]])
    lo, hi = np.zeros(2), np.ones(2)
    none = np.zeros((0, 2, 2))
    assert orc.mc_is_edges(X, [0], [1], 0.05, 500, 3, none, lo, hi)[0] == 0
    lohi = np.array([[[0.45, 0.36], [0.6, 0.6]], [[0.1, 0.7], [0.3, 0.9]]])
    n, sigma = 4000, 0.06
    mc = np.array([orc.mc_edges(X, [0], [1], sigma, n, s, lohi, lo, hi)[0] / n for s in range(30)])
    isv = np.array([float(orc.mc_is_edges(X, [0], [1], sigma, n, s, lohi, lo, hi)[0]) / 2.0 ** 40 / n for s in range(30)])
    se = np.sqrt(mc.var(ddof=1) / 30 + isv.var(ddof=1) / 30)
    assert mc.mean() > 0 and abs(mc.mean() - isv.mean()) < 4 * se, (mc.mean(), isv.mean(), se)
    assert isv.var(ddof=1) * 1.5 < mc.var(ddof=1), (mc.var(ddof=1), isv.var(ddof=1))     # (p ~ 3e-3 here: the gain grows as the event gets rarer)
    # a blocked edge (the midpoint lies inside a box: zero shift): the estimator is plain Monte Carlo, weight 1 per hit
    Xb = np.array([[0.3, 0.5], [0.7, 0.5]])
    box = np.array([[[0.45, 0.4], [0.55, 0.6]]])
    h = orc.mc_edges(Xb, [0], [1], 0.02, 300, 5, box, lo, hi)[0]
    assert orc.mc_is_edges(Xb, [0], [1], 0.02, 300, 5, box, lo, hi)[0] == np.uint64(h) * np.uint64(2 ** 40)


def test_adaptive_importance_sampling_oracle_is_consistent(orc):
    """The scalar loop of the ADAPTIVE estimator (orc_mc_ais_edges): no obstacle -> no shift, nothing collides; a pilot without a collision
    -> plain Monte Carlo (every weight 1); on a rare collision the shift points at the obstacle, the mean agrees with plain Monte Carlo
    over 30 seeds and the variance is several times smaller."""
    X = np.array([[0.2, 0.2], [0.8, 0.25]])
    lo, hi = np.zeros(2), np.ones(2)
    none = np.zeros((0, 2, 2))
    wsum, sh = orc.mc_ais_edges(X, [0], [1], 0.01, 500, 3, none, lo, hi)      # (sigma small enough that no pilot rollout leaves the unit square)
    assert wsum[0] == 0 and np.all(sh == 0)
    lohi = np.array([[[0.45, 0.36], [0.6, 0.6]], [[0.1, 0.7], [0.3, 0.9]]])
    n, sigma = 4000, 0.05
    mc = np.array([orc.mc_edges(X, [0], [1], sigma, n, s, lohi, lo, hi)[0] / n for s in range(30)])
    out = [orc.mc_ais_edges(X, [0], [1], sigma, n, s, lohi, lo, hi) for s in range(30)]
    ais = np.array([float(o[0][0]) / 2.0 ** 40 / n for o in out])
    sh = np.array([o[1][0] for o in out])
    assert np.all(np.abs(sh) <= 3.0) and np.all(sh[:, 1] > 0) and np.all(sh[:, 3] > 0)       # both ends pushed up, towards the box above the segment
    se = np.sqrt(mc.var(ddof=1) / 30 + ais.var(ddof=1) / 30)
    assert mc.mean() > 0 and abs(mc.mean() - ais.mean()) < 4 * se, (mc.mean(), ais.mean(), se)
    assert ais.var(ddof=1) * 3 < mc.var(ddof=1), (mc.var(ddof=1), ais.var(ddof=1))
    # far from every obstacle the pilot sees nothing: zero shift, the weight sum is the plain hit count (zero here)
    far = np.array([[[0.05, 0.9], [0.1, 0.95]]])
    wsum, sh = orc.mc_ais_edges(X, [0], [1], 0.01, 300, 5, far, lo, hi)
    assert wsum[0] == 0 and np.all(sh == 0)
