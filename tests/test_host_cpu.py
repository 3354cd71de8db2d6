"""CPU tests of the host-side pieces that need no GPU: synthetic workloads, radius rule, mirror types."""
import numpy as np

import motionplanning_jl_amd as mp


def test_workloads_are_deterministic_and_shaped_like_baseline():
    a, b = mp.workloads.cfg1(), mp.workloads.cfg1()
    assert np.array_equal(a.X, b.X) and np.array_equal(a.lohi, b.lohi)
    assert (a.N, a.d, a.M) == (1000, 2, 20)
    assert np.array_equal(a.X[0], [0.1, 0.1]) and np.array_equal(a.X[-1], [0.9, 0.9])
    # boxes never contain init or goal
    for p in (a.X[0], a.X[-1]):
        assert not np.any(np.all((a.lohi[:, 0] <= p) & (p <= a.lohi[:, 1]), axis=1))
    w = mp.workloads.cfg2(2000)
    assert (w.d, w.M) == (6, 200) and abs(w.r - mp.workloads.fmt_radius(1.0, 6, 1.0, 2000)) < 1e-15
    assert abs(mp.workloads.fmt_radius(1.0, 6, 1.0, 1_000_000) - 0.17479) < 1e-5
    c4 = mp.workloads.cfg4(500)
    assert c4.X.shape == (500, 4) and np.all(np.abs(c4.X[:, 2:]) <= 0.5)


def test_mirror_types_without_a_device():
    SS = mp.UnitHypercube(3)
    assert mp.dim(SS) == 3 and mp.volume(SS) == 1.0
    DI = mp.DoubleIntegrator(2, vmax=0.5)
    assert mp.dim(DI) == 4 and DI.workspace_dim == 2 and np.allclose(DI.lo, [0, 0, -0.5, -0.5])
    CC = mp.PointRobotNDBoxes([mp.BoxBounds([[0.4, 0.5], [0.19, 0.35]])])          # BoxBounds(lohi::Matrix)
    assert np.allclose(CC.boxes[0].lo, [0.4, 0.19]) and np.allclose(CC.boxes[0].hi, [0.5, 0.35])
    assert CC.lohi().shape == (1, 2, 2) and CC.count == 0
    g = mp.BallGoal([0.9, 0.9], 0.05)
    assert np.allclose(g.params(), [0.9, 0.9, 0.05]) and g.kind == mp._lib.GOAL_BALL
    rng = np.random.default_rng(0)
    s = mp.sample_space(SS, rng, 10)
    assert s.shape == (10, 3) and np.all((0 <= s) & (s <= 1))
    v = g.sample(rng)
    assert np.linalg.norm(v - g.center) <= g.radius


def test_bit_packing_matches_bitvector_chunks():
    bits = np.zeros(200, bool); bits[[0, 63, 64, 199]] = True
    m = mp._lib.pack_bits(bits)
    assert m[0] == (1 | (1 << 63)) and m[1] == 1 and m[3] == (1 << 7)
    assert np.array_equal(mp._lib.unpack_bits(m, 200), bits)
