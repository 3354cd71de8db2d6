"""GPU parity tests (-m gpu): the HIP path, called through the C ABI of include/mpfmt.h, against the CPU
oracle on the same seeded inputs and against the committed golden fixtures.

Bar (BASELINE.json north_star): bit-exact neighbour / collision masks and indices; edge costs within
1e-6 relative (the tests first check them bit-exact and report the tighter result).
"""
import json
import os

import numpy as np
import pytest

import motionplanning_jl_amd as mp

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
COST_RTOL = 1e-6     # north_star tolerance for edge costs


@pytest.fixture(scope="module")
def ctx():
    c = mp.Context(0)
    yield c
    c.close()


def to0(colptr, rowval):
    return colptr - 1, rowval - 1


def graphs_both_paths(ctx, r):
    """Build the r-disc graph through both pair kernels (exact fp64 VALU; fp16 MFMA filter + exact refine)
    and require identical output.  Returns the (1-based) graph and the list of paths that ran."""
    outs, ran = [], []
    for path in (1, 2):
        ctx.set_option("rdisc_path", path)
        try:
            outs.append(ctx.rdisc_graph(r))
            ran.append(ctx.stat("rdisc_path_used"))
        except mp.MPFMTError as e:
            assert path == 2 and e.code == mp._lib.ERR_ARG      # filter not usable for this (d, r)
    ctx.set_option("rdisc_path", 0)
    for o in outs[1:]:
        for x, y in zip(outs[0], o):
            assert np.array_equal(x, y)
    return outs[0], ran


def check_costs(got, want):
    assert got.shape == want.shape
    if want.size:
        rel = np.max(np.abs(got - want) / np.maximum(np.abs(want), 1e-300))
        assert rel <= COST_RTOL, rel


# ---- golden fixtures -------------------------------------------------------------------------------

@pytest.mark.parametrize("name", ["BOXES2D", "BOXES3D"])
def test_golden_segments(ctx, name):
    z = np.load(os.path.join(G, "segments_%s.npz" % name))
    P, Q = z["P"], z["Q"]
    ctx.upload_boxes(z["lohi"], z["ss_lo"], z["ss_hi"])
    got = mp._lib.unpack_bits(ctx.motions_free(P, Q), len(P))
    assert np.array_equal(got, z["free_full"])
    ctx.upload_boxes(z["lohi"])                          # no state-space bounds: the box predicate alone
    got = mp._lib.unpack_bits(ctx.motions_free(P, Q), len(P))
    assert np.array_equal(got, z["free_boxes"])
    ctx.upload_boxes(z["lohi"], z["ss_lo"], z["ss_hi"])
    got = mp._lib.unpack_bits(ctx.states_free(P), len(P))
    assert np.array_equal(got, z["point_free"])


def test_golden_known_answers(ctx):
    fx = json.load(open(os.path.join(G, "boxes_nd.json")))
    lo, hi = fx["BOXES2D"][1]
    cases = fx["known_answers_BOXES2D_box2"]["cases"]
    ctx.upload_boxes(np.array([[lo, hi]]))
    P = np.array([c[0] for c in cases]); Q = np.array([c[1] for c in cases])
    got = mp._lib.unpack_bits(ctx.motions_free(P, Q), len(P))
    assert got.tolist() == [c[4] for c in cases]


@pytest.mark.parametrize("tag", ["d2_n1000", "d6_n1500"])
def test_golden_rdisc(ctx, tag):
    z = np.load(os.path.join(G, "rdisc_%s.npz" % tag))
    ctx.upload_samples(z["X"])
    (colptr, rowval, nzval), ran = graphs_both_paths(ctx, float(z["r"]))
    assert ran == [1, 2]
    c0, r0 = to0(colptr, rowval)
    assert np.array_equal(c0, z["colptr"])
    assert np.array_equal(r0, z["rowval"])
    check_costs(nzval, z["nzval"])


def test_golden_fmt_cfg1(ctx):
    z = np.load(os.path.join(G, "fmt_cfg1.npz"))
    ctx.upload_samples(z["X"])
    ctx.upload_boxes(z["lohi"], np.zeros(2), np.ones(2))
    res = ctx.fmtstar(float(z["r"]), mp._lib.GOAL_BALL, z["goal"], init_idx=1, checkpts=True)
    assert res["status"] == 1 == int(z["status"])
    assert res["collision_checks"] == int(z["collision_checks"])
    assert np.array_equal(res["A"] - 1, z["A"])
    assert np.array_equal(res["path"] - 1, z["path"])
    assert res["z"] - 1 == int(z["z"])
    check_costs(res["C"], z["C"])
    assert abs(res["cost"] - float(z["cost"])) <= COST_RTOL * float(z["cost"])


# ---- seeded random parity -------------------------------------------------------------------------------

@pytest.mark.parametrize("N,d,r", [(5000, 2, 0.03), (4000, 3, 0.09), (6000, 4, 0.16), (20000, 6, 0.33),
                                   (3000, 12, 0.95), (777, 1, 0.01), (2500, 5, 0.3), (1500, 16, 1.3)])
def test_rdisc_random(ctx, orc, N, d, r):
    rng = np.random.default_rng(100 + d)
    X = rng.random((N, d))
    ctx.upload_samples(X)
    (colptr, rowval, nzval), ran = graphs_both_paths(ctx, r)
    assert (2 in ran) == (d <= 12)
    oc, orow, oval = orc.rdisc_graph(X, r)
    c0, r0 = to0(colptr, rowval)
    assert np.array_equal(c0, oc)
    assert np.array_equal(r0, orow)
    check_costs(nzval, oval)


def test_sqrt_is_correctly_rounded(ctx, orc):
    """Edge costs are sqrt(d2); if the device sqrt is IEEE-correct the costs are bit-identical to the oracle."""
    rng = np.random.default_rng(3)
    X = rng.random((8000, 6))
    ctx.upload_samples(X)
    _, _, nzval = ctx.rdisc_graph(0.4)
    _, _, oval = orc.rdisc_graph(X, 0.4)
    assert np.array_equal(nzval, oval)


def test_rdisc_clustered_and_duplicates(ctx, orc):
    rng = np.random.default_rng(8)
    X = np.concatenate([0.5 + 0.01 * rng.standard_normal((1500, 3)), rng.random((1500, 3)),
                        np.tile(rng.random((1, 3)), (40, 1))])          # dense cluster + 40 identical points
    rng.shuffle(X)
    ctx.upload_samples(X)
    for r in (0.0, 0.02, 0.25):
        (colptr, rowval, nzval), _ = graphs_both_paths(ctx, r)
        oc, orow, oval = orc.rdisc_graph(X, r)
        c0, r0 = to0(colptr, rowval)
        assert np.array_equal(c0, oc) and np.array_equal(r0, orow)
        check_costs(nzval, oval)


def test_rdisc_pool_overflow_falls_back(orc):
    """A dense cluster defeats the pool's capacity estimate: the build must notice the overflow and run the fill pass."""
    rng = np.random.default_rng(81)
    X = np.concatenate([0.5 + 0.004 * rng.standard_normal((1800, 3)), rng.random((500, 3))])      # (every pair of the cluster is an edge: columns of 1800+, under the ordering kernel's 2048)
    c = mp.Context(0)
    c.upload_samples(X)
    used = []
    for _ in range(2):                                    # second build has the capacity hint of the first
        colptr, rowval, nzval = c.rdisc_graph(0.2)
        used.append(c.stat("pool_used"))
    c.set_option("rdisc_pool", 0)
    c2, r2, n2 = c.rdisc_graph(0.2)
    assert np.array_equal(colptr, c2) and np.array_equal(rowval, r2) and np.array_equal(nzval, n2)
    oc, orow, oval = orc.rdisc_graph(X, 0.2)
    assert np.array_equal(colptr - 1, oc) and np.array_equal(rowval - 1, orow) and np.array_equal(nzval, oval)
    assert used == [0, 1]
    c.close()


@pytest.mark.parametrize("N", [1, 2, 63, 64, 65, 129])
def test_rdisc_tiny_and_ragged(ctx, orc, N):
    rng = np.random.default_rng(N)
    X = rng.random((N, 3))
    ctx.upload_samples(X)
    for r in (0.3, 5.0):                  # r = 5: every pair is a neighbour (brute-force regime)
        (colptr, rowval, nzval), _ = graphs_both_paths(ctx, r)
        oc, orow, oval = orc.rdisc_graph(X, r)
        c0, r0 = to0(colptr, rowval)
        assert np.array_equal(c0, oc) and np.array_equal(r0, orow)
        check_costs(nzval, oval)


@pytest.mark.parametrize("N", [1, 2, 15, 16, 17, 63, 64, 65, 129])
def test_graph_sweep_tiny_and_ragged(ctx, orc, N):
    """Graph sweep on fewer columns than one task / one wavefront, empty columns, all-pairs degree > 64 (several rounds
    per column), with and without obstacles."""
    rng = np.random.default_rng(500 + N)
    X = rng.random((N, 3))
    lohi = mp.workloads.make_boxes(rng, 7, 3, 0.05, 0.2, [])
    ss_lo, ss_hi = np.full(3, 0.05), np.full(3, 0.95)
    ctx.upload_samples(X)
    for boxes in (lohi, lohi[:0]):
        ctx.upload_boxes(boxes, ss_lo, ss_hi)
        for r in (0.25, 5.0):
            colptr, rowval, _ = ctx.rdisc_graph(r)
            got = ctx.graph_edges_free()
            c0, r0 = to0(colptr, rowval)
            want = orc.graph_edges_free(X, c0, r0, boxes, ss_lo, ss_hi)
            assert np.array_equal(got, want)


def test_graph_sweep_long_columns_fill_the_queue(ctx, orc):
    """Dense graph in a cluttered world: most entries fail some broad phase, so the narrow-phase queue runs full
    passes, flushes at task boundaries, overflows into the in-place path, and columns span many rounds."""
    rng = np.random.default_rng(77)
    N, d = 700, 2
    X = rng.random((N, d))
    lohi = mp.workloads.make_boxes(rng, 150, d, 0.01, 0.04, [])
    ctx.upload_samples(X)
    ctx.upload_boxes(lohi, np.zeros(d), np.ones(d))
    colptr, rowval, _ = ctx.rdisc_graph(0.6)
    got = ctx.graph_edges_free()
    c0, r0 = to0(colptr, rowval)
    assert int(np.max(np.diff(c0))) > 300
    want = orc.graph_edges_free(X, c0, r0, lohi, np.zeros(d), np.ones(d))
    assert np.array_equal(got, want)
    assert 0.02 < mp._lib.unpack_bits(got, len(r0)).mean() < 0.98


def test_rdisc_high_degree_columns(ctx, orc):
    """Columns longer than the LDS-resident sort window (exercises the chunked rank path)."""
    rng = np.random.default_rng(17)
    X = rng.random((3000, 2))
    ctx.upload_samples(X)
    (colptr, rowval, nzval), ran = graphs_both_paths(ctx, 1.2)
    assert ran == [1, 2]
    oc, orow, oval = orc.rdisc_graph(X, 1.2)
    c0, r0 = to0(colptr, rowval)
    assert int(np.max(np.diff(oc))) > 2048
    assert np.array_equal(c0, oc) and np.array_equal(r0, orow)
    check_costs(nzval, oval)


def test_rdisc_query(ctx, orc):
    rng = np.random.default_rng(9)
    X = rng.random((5000, 4))
    ctx.upload_samples(X)
    for v in (1, 2, 2500, 5000):
        inds, ds = ctx.rdisc_query(v, 0.2)
        oi, od = orc.inball(X, v - 1, 0.2)
        assert np.array_equal(inds - 1, oi)
        check_costs(ds, od)
    with pytest.raises(mp.MPFMTError) as e:
        ctx.rdisc_query(1, 0.2, cap=1)
    assert e.value.code == mp._lib.ERR_CAPACITY


def random_world(rng, N, d, M, h_lo, h_hi):
    X = rng.random((N, d))
    c = rng.random((M, d)); h = h_lo + (h_hi - h_lo) * rng.random((M, d))
    lohi = np.stack([c - h, c + h], axis=1)
    return X, lohi


@pytest.mark.parametrize("d,M,h", [(2, 20, (0.02, 0.08)), (3, 10, (0.05, 0.2)), (6, 200, (0.1, 0.2)), (6, 0, (0.1, 0.2)),
                                   (6, 700, (0.05, 0.15)), (12, 64, (0.2, 0.35))])
def test_edges_and_points_random(ctx, orc, d, M, h):
    rng = np.random.default_rng(200 + d + M)
    N = 20000
    X, lohi = random_world(rng, N, d, M, *h)
    X[:50] = X[:50] * 1.2 - 0.1               # some samples outside the state-space bounds
    ss_lo, ss_hi = np.zeros(d), np.ones(d)
    ctx.upload_samples(X)
    ctx.upload_boxes(lohi, ss_lo, ss_hi, dw=d)
    assert np.array_equal(ctx.points_free(), orc.points_free(X, lohi, ss_lo, ss_hi))
    idx = rng.integers(1, N + 1, size=3001)
    assert np.array_equal(ctx.points_free(idx), orc.points_free(X, lohi, ss_lo, ss_hi, idx=idx - 1))
    E = 100003
    src = rng.integers(1, N + 1, size=E)
    near = X[src - 1] + 0.1 * (rng.random((E, d)) - 0.5)          # FMT*-like short edges: snap to nearest-ish sample
    dst = rng.integers(1, N + 1, size=E)
    dst[:200] = src[:200]                                          # zero-length edges
    got = ctx.edges_free(src, dst)
    want = orc.edges_free(X, src - 1, dst - 1, lohi, ss_lo, ss_hi)
    assert np.array_equal(got, want)
    # explicit short segments
    got = ctx.motions_free(X[src - 1], near)
    want = orc.pack(np.array([orc.is_free_motion(a, b, lohi, ss_lo, ss_hi) for a, b in zip(X[src[:4000] - 1], near[:4000])]))
    assert np.array_equal(mp._lib.unpack_bits(got, E)[:4000], orc.unpack(want, 4000))


@pytest.mark.parametrize("N,d,M,r", [(1000, 2, 20, 0.0663), (6000, 6, 200, 0.4), (4000, 3, 300, 0.12),
                                     (3000, 8, 100, 0.5), (2000, 12, 50, 0.7), (5000, 6, 600, 0.35)])
def test_graph_edges_free(ctx, orc, N, d, M, r):
    rng = np.random.default_rng(300 + d)
    X, lohi = random_world(rng, N, d, M, 0.05, 0.15)
    ss_lo, ss_hi = np.full(d, 0.02), np.full(d, 0.98)
    ctx.upload_samples(X)
    ctx.upload_boxes(lohi, ss_lo, ss_hi)
    colptr, rowval, _ = ctx.rdisc_graph(r)
    got = ctx.graph_edges_free()
    c0, r0 = to0(colptr, rowval)
    want = orc.graph_edges_free(X, c0, r0, lohi, ss_lo, ss_hi)
    assert np.array_equal(got, want)
    # the same bits through the explicit edge list entry point (rows are parents, columns children)
    cols = np.repeat(np.arange(1, N + 1), np.diff(colptr))
    assert np.array_equal(ctx.edges_free(rowval, cols), want)


def test_expand_step(ctx, orc):
    rng = np.random.default_rng(41)
    N, d, r = 5000, 3, 0.1
    X, lohi = random_world(rng, N, d, 30, 0.05, 0.15)
    ss_lo, ss_hi = np.zeros(d), np.ones(d)
    ctx.upload_samples(X)
    ctx.upload_boxes(lohi, ss_lo, ss_hi)
    ctx.rdisc_graph(r)
    F = ctx.points_free()
    H = rng.random(N) < 0.1
    W = ~H & (rng.random(N) < 0.8)
    Cc = np.where(H, rng.random(N), 0.0)
    zs = np.flatnonzero(H)[:40] + 1
    for swept in (False, True):
        if swept:
            ctx.graph_edges_free()
        xs, ym, cm, fr = ctx.expand(mp._lib.pack_bits(W), mp._lib.pack_bits(H), F, Cc, zs)
        oxs, oym, ocm, ofr = orc.expand(X, r, orc.pack(W), orc.pack(H), F, Cc, zs - 1, lohi, ss_lo, ss_hi)
        assert np.array_equal(xs - 1, oxs) and np.array_equal(ym - 1, oym)
        check_costs(cm, ocm)
        assert np.array_equal(fr, ofr)


@pytest.mark.parametrize("N,d,M,seed", [(3000, 2, 25, 1), (5000, 3, 40, 2), (8000, 6, 100, 3)])
def test_fmtstar_matches_oracle(ctx, orc, N, d, M, seed):
    w = mp.workloads.make("t", N, d, M, 0.05, 0.12, seed=seed, goal_radius=0.1)
    ctx.upload_samples(w.X)
    ctx.upload_boxes(w.lohi, w.ss_lo, w.ss_hi)
    res = ctx.fmtstar(w.r, mp._lib.GOAL_BALL, w.goal_params())
    ref = orc.fmtstar(w.X, w.r, orc.GOAL_BALL, w.goal_params(), w.lohi, w.ss_lo, w.ss_hi, nn_mode=1)
    assert res["status"] == ref["status"]
    assert res["collision_checks"] == ref["collision_checks"]
    assert np.array_equal(res["A"] - 1, ref["A"])
    assert np.array_equal(res["path"] - 1, ref["path"])
    check_costs(res["C"], ref["C"])


def test_fmtstar_infeasible_init(ctx):
    X = np.array([[0.1, 0.1], [0.5, 0.5], [0.9, 0.9]])
    ctx.upload_samples(X)
    ctx.upload_boxes(np.array([[[0.0, 0.0], [0.2, 0.2]]]), np.zeros(2), np.ones(2))
    with pytest.raises(mp.MPFMTError) as e:
        ctx.fmtstar(0.7, mp._lib.GOAL_BALL, [0.9, 0.9, 0.05])
    assert e.value.code == mp._lib.ERR_INFEASIBLE


def test_error_behaviour(ctx):
    ctx.upload_samples(np.random.default_rng(0).random((100, 3)))
    with pytest.raises(mp.MPFMTError) as e:
        ctx.rdisc_query(0, 0.1)
    assert e.value.code == mp._lib.ERR_ARG
    with pytest.raises(mp.MPFMTError) as e:
        ctx.edges_free([1], [101])
    assert e.value.code == mp._lib.ERR_ARG
    c2 = mp.Context(0)
    c2.upload_samples(np.zeros((4, 2)))
    with pytest.raises(mp.MPFMTError) as e:
        c2.points_free()
    assert e.value.code == mp._lib.ERR_STATE
    c2.close()


def test_determinism(ctx):
    w = mp.workloads.make("t", 30000, 6, 200, 0.1, 0.2, seed=5)
    outs = []
    for _ in range(2):
        ctx.upload_samples(w.X)
        ctx.upload_boxes(w.lohi, w.ss_lo, w.ss_hi)
        colptr, rowval, nzval = ctx.rdisc_graph(0.3)
        outs.append((colptr, rowval, nzval, ctx.graph_edges_free()))
    for a, b in zip(*outs):
        assert np.array_equal(a, b)


def test_shards_partition_the_graph(ctx, orc):
    """Two shards on one GPU: their column sets are disjoint and their union is the full graph."""
    rng = np.random.default_rng(77)
    X = rng.random((9000, 6))
    r = 0.35
    oc, orow, _ = orc.rdisc_graph(X, r)
    deg = np.zeros(9000, dtype=np.int64)
    rows_by_col = {}
    for rank in range(2):
        c = mp.Context(0)
        c.set_shard(rank, 2)
        c.upload_samples(X)
        (colptr, rowval, _), ran = graphs_both_paths(c, r)
        assert ran == [1, 2]
        k = np.diff(colptr)
        assert np.all((deg == 0) | (k == 0))
        for v in np.flatnonzero(k):
            rows_by_col[v] = rowval[colptr[v] - 1:colptr[v + 1] - 1] - 1
        deg += k
        c.close()
    assert np.array_equal(deg, np.diff(oc))
    for v in range(0, 9000, 41):
        assert np.array_equal(rows_by_col.get(v, np.zeros(0, np.int64)), orow[oc[v]:oc[v + 1]])


@pytest.mark.parametrize("world", [2, 3, 8])
def test_sharded_sweep_masks(orc, world):
    """Each shard sweeps only its own columns (visited in cell-sorted order through perm): the mask over its local CSC
    equals the oracle's on the same local graph, and the shards' free-edge counts add up to the unsharded total."""
    rng = np.random.default_rng(123)
    N, d, r = 12000, 3, 0.11
    X, lohi = random_world(rng, N, d, 80, 0.04, 0.12)
    lo, hi = np.full(d, 0.03), np.full(d, 0.97)
    full = mp.Context(0)
    full.upload_samples(X); full.upload_boxes(lohi, lo, hi)
    colptr, rowval, _ = full.rdisc_graph(r)
    total_free = int(mp._lib.unpack_bits(full.graph_edges_free(), len(rowval)).sum())
    full.close()
    got_free = 0
    for rank in range(world):
        c = mp.Context(0)
        c.set_shard(rank, world)
        c.upload_samples(X); c.upload_boxes(lohi, lo, hi)
        cp, rv, _ = c.rdisc_graph(r)
        m = c.graph_edges_free()
        want = orc.graph_edges_free(X, cp - 1, rv - 1, lohi, lo, hi)
        assert np.array_equal(m, want)
        got_free += int(mp._lib.unpack_bits(m, len(rv)).sum())
        c.close()
    assert got_free == total_free


# ---- full-size properties (BASELINE.json configs[1]: R^6, N=100k, M=200) ---------------------------------

def test_cfg2_full_size_properties(ctx, orc):
    w = mp.workloads.cfg2()
    N = w.N
    ctx.upload_samples(w.X)
    ctx.upload_boxes(w.lohi, w.ss_lo, w.ss_hi)
    colptr, rowval, nzval = ctx.rdisc_graph(w.r)
    assert ctx.stat("rdisc_path_used") == 2               # the shipped (auto) path is the MFMA filter
    assert ctx.stat("pool_used") == 1                     # ... in its single-pass form (hits pooled by the count pass)
    c0, r0 = to0(colptr, rowval)
    nnz = len(rowval)
    assert c0[0] == 0 and c0[-1] == nnz and nnz % 2 == 0
    cols = np.repeat(np.arange(N), np.diff(c0))
    assert np.all(r0 != cols)                                             # self excluded
    same_col = cols[1:] == cols[:-1]
    assert np.all(np.diff(r0)[same_col] > 0)                              # ascending inside every column
    assert np.all(nzval <= w.r * (1 + 1e-12))
    # symmetry of the metric graph: the multiset of (i,j) equals the multiset of (j,i)
    k1 = np.sort(cols.astype(np.int64) * N + r0)
    k2 = np.sort(r0.astype(np.int64) * N + cols)
    assert np.array_equal(k1, k2)
    # sampled columns against the oracle KD-tree
    kd = orc.KDTree(w.X)
    rng = np.random.default_rng(1)
    for v in rng.integers(0, N, size=300):
        oi, od = kd.inball(int(v), w.r)
        assert np.array_equal(r0[c0[v]:c0[v + 1]], oi)
        check_costs(nzval[c0[v]:c0[v + 1]], od)
    # edge mask: sampled entries against the oracle, plus the point mask in full
    mask = mp._lib.unpack_bits(ctx.graph_edges_free(), nnz)
    es = rng.integers(0, nnz, size=200000)
    want = orc.unpack(orc.edges_free(w.X, r0[es], cols[es], w.lohi, w.ss_lo, w.ss_hi), len(es))
    assert np.array_equal(mask[es], want)
    assert np.array_equal(ctx.points_free(), orc.points_free(w.X, w.lohi, w.ss_lo, w.ss_hi))


# ---- double integrator (BASELINE.json configs[3]) ------------------------------------------

def di_world(N, seed, M=20):
    rng = np.random.default_rng(seed)
    X = np.concatenate([rng.random((N, 2)), rng.random((N, 2)) - 0.5], axis=1)
    X[0] = [0.1, 0.1, 0.0, 0.0]
    X[-1] = [0.9, 0.9, 0.0, 0.0]
    X[5] = X[3]                                     # duplicate state (steer returns (0, 0))
    lohi = mp.workloads.make_boxes(rng, M, 2, 0.02, 0.08, [X[0, :2], X[-1, :2]])
    ss_lo = np.array([0.0, 0.0, -0.5, -0.5]); ss_hi = np.array([1.0, 1.0, 0.5, 0.5])
    return X, lohi, ss_lo, ss_hi


def test_di_steer_batch(ctx, orc):
    rng = np.random.default_rng(31)
    n = 4096
    X0 = np.concatenate([rng.random((n, 2)), rng.random((n, 2)) - 0.5], axis=1)
    X1 = X0 + np.concatenate([0.3 * (rng.random((n, 2)) - 0.5), 0.4 * (rng.random((n, 2)) - 0.5)], axis=1)
    X1[:7] = X0[:7]                                 # identical states
    for rho, r in ((1.0, 1.0), (0.3, 0.7)):
        cost, t = ctx.di_steer(X0, X1, rho, r)
        ref = np.array([orc.di_steer(a, b, rho, r) for a, b in zip(X0, X1)])
        assert np.array_equal(cost, ref[:, 0]) and np.array_equal(t, ref[:, 1])      # identical operation sequences
    # 3-D workspace (6-D state)
    X0 = np.concatenate([rng.random((512, 3)), rng.random((512, 3)) - 0.5], axis=1)
    X1 = np.concatenate([rng.random((512, 3)), rng.random((512, 3)) - 0.5], axis=1)
    cost, t = ctx.di_steer(X0, X1, 1.0, 1.5)
    ref = np.array([orc.di_steer(a, b, 1.0, 1.5) for a, b in zip(X0, X1)])
    check_costs(cost, ref[:, 0]); check_costs(t, ref[:, 1])


@pytest.mark.parametrize("N,rho,r", [(1500, 1.0, 1.0), (900, 0.5, 0.6), (130, 1.0, 3.0)])
def test_di_graph_and_sweep(ctx, orc, N, rho, r):
    X, lohi, ss_lo, ss_hi = di_world(N, 40 + N)
    ctx.upload_samples(X)
    ctx.upload_boxes(lohi, ss_lo, ss_hi)
    colptr, rowval, nzval, tval = ctx.di_graph(rho, r)
    oc, orow, oval, otv = orc.di_pairwise(X, rho, r)
    assert np.array_equal(colptr - 1, oc) and np.array_equal(rowval - 1, orow)
    check_costs(nzval, oval); check_costs(tval, otv)
    assert np.array_equal(nzval, oval) and np.array_equal(tval, otv)
    mask, nseg = ctx.di_graph_edges_free()
    assert np.array_equal(mask, orc.di_graph_edges_free(X, rho, r, oc, orow, lohi, ss_lo, ss_hi))
    assert nseg.max() <= 4


@pytest.mark.parametrize("m,N,rho,r,scale,offset", [(2, 3000, 1.0, 0.8, 1.0, 0.0), (2, 5000, 0.5, 1.1, 1.0, 0.0), (2, 4500, 2.0, 0.6, 1.0, 0.0),
                                                 (2, 3000, 1.0, 0.9, 1.0, 50.0), (2, 3000, 1.0, 5.0, 7.0, -3.0), (1, 2500, 1.0, 0.7, 1.0, 0.0),
                                                 (1, 2000, 3.0, 0.6, 1.0, 10.0), (2, 70, 1.0, 1.0, 1.0, 0.0), (2, 4097, 1.0, 0.25, 1.0, 0.0)])
def test_di_matrix_core_prefilter_equals_the_vector_alu_test(orc, m, N, rho, r, scale, offset):
    """The double-integrator build with its candidate test as an fp16 bilinear form on the matrix cores (option di_path = 2) against the
    vector-ALU form (1) and the oracle: the same graph, costs and optimal times bit for bit -- and the same NUMBER of pairs reaching the
    Newton iteration (stat `survivors`): both forms apply the identical fp64 tests behind their filter, so a pair the fp16 filter wrongly
    dropped would show as a smaller count before it showed as a missing edge.  Radii small and large against the samples' extent, positions
    far from the origin (the form is centred), scaled workspaces, other rho, one-dimensional workspaces, a set smaller than two tiles."""
    rng = np.random.default_rng(300 + N + m)
    vmax = 0.5 * scale
    X = np.concatenate([offset + scale * rng.random((N, m)), vmax * (2 * rng.random((N, m)) - 1)], axis=1)
    X[: N // 50, m:] = 0.0                                                # some states at rest
    X[N // 50: N // 50 + 5] = X[:5]                                       # repeated states (steer returns (0, 0): never an edge)
    got = {}
    for path in (1, 2):
        with mp.Context(0) as c:
            c.set_option("di_path", 1 if path == 1 else 0)                # (0 = auto: the matrix cores wherever the error bound allows)
            c.upload_samples(X)
            got[path] = c.di_graph(rho, r) + (c.stat("survivors"), c.stat("di_path_used"))
    assert got[1][5] == 1
    # (r = 0.25 in a unit workspace: the scaled features reach ~100 and the fp16 bound is no longer small against 1 / rho -- the library must
    # say so and keep the vector-ALU test; everywhere else the matrix-core form must have run)
    assert got[2][5] == (1 if r == 0.25 else 2), got[2][5]
    if r == 0.25:
        with mp.Context(0) as c:
            c.set_option("di_path", 2)
            c.upload_samples(X)
            with pytest.raises(Exception):
                c.di_graph(rho, r)
    for u, v in zip(got[1][:4], got[2][:4]):
        assert np.array_equal(u, v)
    assert got[1][4] == got[2][4], (got[1][4], got[2][4])
    oc, orow, oval, otv = orc.di_pairwise(X, rho, r)
    colptr, rowval, nzval, tval = got[2][:4]
    assert np.array_equal(colptr - 1, oc) and np.array_equal(rowval - 1, orow)
    assert np.array_equal(nzval, oval) and np.array_equal(tval, otv)
    assert len(orow) > 0


def test_di_single_pass_build(orc):
    """N >= 4096 takes the single-pass build (hits kept in slot lists sized by a pilot); it must equal both the oracle and the
    two-pass build, and a capacity overflow must fall back cleanly."""
    N = 4300
    X, lohi, ss_lo, ss_hi = di_world(N, 5)
    oc, orow, oval, otv = orc.di_pairwise(X, 1.0, 0.7)
    outs = []
    for pool in (1, 0):
        c = mp.Context(0)
        c.set_option("rdisc_pool", pool)
        c.upload_samples(X); c.upload_boxes(lohi, ss_lo, ss_hi)
        c.timing_reset()
        outs.append(c.di_graph(1.0, 0.7))
        assert (c.timing("di_fill")[0] > 0)
        c.close()
    for a, b in zip(outs[0], outs[1]):
        assert np.array_equal(a, b)
    colptr, rowval, nzval, tval = outs[0]
    assert np.array_equal(colptr - 1, oc) and np.array_equal(rowval - 1, orow)
    assert np.array_equal(nzval, oval) and np.array_equal(tval, otv)
    # a cluster the pilot does not see (only every 32nd tile is sampled): the tail tiles are far denser -> overflow -> fallback
    X2 = X.copy()
    X2[-200:, :2] = 0.5 + 0.002 * np.random.default_rng(1).standard_normal((200, 2)); X2[-200:, 2:] = 0.0
    c = mp.Context(0)
    c.upload_samples(X2); c.upload_boxes(lohi, ss_lo, ss_hi)
    cp, rv, nz, tv = c.di_graph(1.0, 0.7)
    o2 = orc.di_pairwise(X2, 1.0, 0.7)
    assert np.array_equal(cp - 1, o2[0]) and np.array_equal(rv - 1, o2[1]) and np.array_equal(nz, o2[2])
    c.close()


def test_di_fmtstar_matches_oracle(ctx, orc):
    X, lohi, ss_lo, ss_hi = di_world(2500, 77)
    ctx.upload_samples(X)
    ctx.upload_boxes(lohi, ss_lo, ss_hi)
    res = ctx.di_fmtstar(1.0, 1.0, mp._lib.GOAL_POINT, X[-1])
    oc, orow, oval, _ = orc.di_pairwise(X, 1.0, 1.0)
    ref = orc.di_fmtstar(X, 1.0, 1.0, oc, orow, oval, orc.GOAL_POINT, X[-1], lohi, ss_lo, ss_hi)
    assert res["status"] == ref["status"] == 1
    assert res["collision_checks"] == ref["collision_checks"]
    assert np.array_equal(res["A"] - 1, ref["A"]) and np.array_equal(res["path"] - 1, ref["path"])
    check_costs(res["C"], ref["C"])
    # workspace ball goal
    res = ctx.di_fmtstar(1.0, 1.0, mp._lib.GOAL_BALL, [0.9, 0.9, 0.1])
    ref = orc.di_fmtstar(X, 1.0, 1.0, oc, orow, oval, orc.GOAL_BALL, [0.9, 0.9, 0.1], lohi, ss_lo, ss_hi)
    assert res["status"] == ref["status"] and np.array_equal(res["A"] - 1, ref["A"])
    assert res["collision_checks"] == ref["collision_checks"]


# ---- full size: the configuration BASELINE.json's metric is quoted on (R^6, N=1e6, M=200) -----------------

def test_north_star_full_size_properties(orc):
    """Size-independent properties at N=1e6 (the oracle cannot brute-force this size): column contract, symmetry of
    the metric graph (checksum of the (i,j) and (j,i) key multisets), sampled columns against the oracle's KD-tree,
    sampled edge bits and the whole point mask against the oracle."""
    w = mp.workloads.north_star()
    N = w.N
    ctx = mp.Context(0)
    ctx.upload_samples(w.X)
    ctx.upload_boxes(w.lohi, w.ss_lo, w.ss_hi)
    colptr, rowval, nzval = ctx.rdisc_graph(w.r)
    assert ctx.stat("rdisc_path_used") == 2
    nnz = len(rowval)
    deg = np.diff(colptr)
    assert colptr[0] == 1 and colptr[-1] == nnz + 1 and nnz % 2 == 0 and deg.min() >= 0
    cols = np.repeat(np.arange(1, N + 1, dtype=np.int64), deg)
    assert not np.any(rowval == cols)                                            # self excluded
    d = np.diff(rowval)
    same = cols[1:] == cols[:-1]
    assert np.all(d[same] > 0)                                                   # ascending inside every column
    del d, same
    assert nzval.max() <= w.r * (1 + 1e-12) and nzval.min() > 0
    # symmetry: sum and xor checksums of the directed keys and of their transposes agree
    k1 = cols * N + rowval
    k2 = rowval * N + cols
    assert int(k1.sum()) == int(k2.sum()) and int(np.bitwise_xor.reduce(k1)) == int(np.bitwise_xor.reduce(k2))
    del k1, k2
    # every undirected edge carries the same cost in both directions: sum of nzval over (i<j) == over (i>j)
    up = rowval < cols
    assert abs(float(nzval[up].sum()) - float(nzval[~up].sum())) <= 1e-9 * float(nzval.sum())
    kd = orc.KDTree(w.X)
    rng = np.random.default_rng(2)
    for v in rng.integers(0, N, size=400):
        oi, od = kd.inball(int(v), w.r)
        a, b = colptr[v] - 1, colptr[v + 1] - 1
        assert np.array_equal(rowval[a:b] - 1, oi)
        assert np.array_equal(nzval[a:b], od)
    mask = mp._lib.unpack_bits(ctx.graph_edges_free(), nnz)
    es = rng.integers(0, nnz, size=300000)
    want = orc.unpack(orc.edges_free(w.X, rowval[es] - 1, cols[es] - 1, w.lohi, w.ss_lo, w.ss_hi), len(es))
    assert np.array_equal(mask[es], want)
    assert np.array_equal(ctx.points_free(), orc.points_free(w.X, w.lohi, w.ss_lo, w.ss_hi))
    ctx.close()


# ---- batch free-space sampler ----------------------------------------------------------------------------

@pytest.mark.parametrize("N,d,M,goal_kind,seed", [(500, 2, 20, 1, 1), (3000, 3, 60, 0, 2), (20000, 6, 200, 1, 3), (70001, 2, 150, 2, 4)])
def test_sample_free_matches_oracle(ctx, orc, N, d, M, goal_kind, seed):
    """Device rejection sampler vs the scalar loop on the same counter-based stream: identical samples (bit for bit),
    identical number of candidates consumed; batches of any size reproduce the sequential order."""
    rng = np.random.default_rng(900 + seed)
    init, gc = np.full(d, 0.1), np.full(d, 0.9)
    lohi = mp.workloads.make_boxes(rng, M, d, 0.03, 0.12, [init, gc])
    lo, hi = np.full(d, -0.25), np.full(d, 1.5)
    goal = {0: np.concatenate([gc - 0.05, gc + 0.05]), 1: np.concatenate([gc, [0.08]]), 2: gc}[goal_kind]
    ctx.upload_boxes(lohi, lo, hi)
    X, att = ctx.sample_free(seed, N, init=init, goal_kind=goal_kind, goal_params=goal, goal_ct=5)
    rc, W, oatt = orc.sample_free(seed, N, d, init, lohi, lo, hi, goal_kind, goal, goal_ct=5)
    assert rc == 0
    assert att == oatt
    assert np.array_equal(X, W)
    # the set is live in the context: the checkpts bitmap over it is all ones and a graph can be built straight away
    assert mp._lib.unpack_bits(ctx.points_free(), N).all()
    colptr, rowval, _ = ctx.rdisc_graph(0.05)
    assert colptr[-1] - 1 == len(rowval)


def test_sample_free_without_init_or_goal_and_errors(ctx, orc):
    d = 4
    rng = np.random.default_rng(12)
    lohi = mp.workloads.make_boxes(rng, 40, d, 0.1, 0.25, [])
    ctx.upload_boxes(lohi, np.zeros(d), np.ones(d))
    X, att = ctx.sample_free(99, 5000)
    rc, W, oatt = orc.sample_free(99, 5000, d, None, lohi, np.zeros(d), np.ones(d), 0, np.zeros(2 * d), goal_ct=0)
    assert np.array_equal(X, W) and att == oatt and att > 5000          # some candidates were rejected
    ctx.upload_boxes(lohi)                                               # no bounds -> nothing to sample from
    with pytest.raises(mp.MPFMTError) as e:
        ctx.sample_free(1, 10)
    assert e.value.code == mp._lib.ERR_STATE
    ctx.upload_boxes(np.array([[np.zeros(d) - 1, np.ones(d) + 1]]), np.zeros(d), np.ones(d))   # one box covers everything
    with pytest.raises(mp.MPFMTError) as e:
        ctx.sample_free(1, 10)
    assert e.value.code == mp._lib.ERR_INFEASIBLE


def test_sample_free_north_star_size_properties(orc):
    """N = 1e6 in R^6 with 200 boxes: size-independent properties + spot parity of a prefix against the scalar loop."""
    w = mp.workloads.north_star()
    c = mp.Context(0)
    c.upload_boxes(w.lohi, w.ss_lo, w.ss_hi)
    X, att = c.sample_free(2024, w.N, init=w.init, goal_kind=mp._lib.GOAL_BALL, goal_params=w.goal_params(), goal_ct=5)
    assert X.shape == (w.N, 6) and att >= w.N - 1
    assert np.array_equal(X[0], w.init)
    assert np.all((X >= 0) & (X <= 1))
    assert np.all(np.linalg.norm(X[-5:] - w.goal_center, axis=1) <= w.goal_radius)
    assert mp._lib.unpack_bits(c.points_free(), w.N).all()                 # every sample is a free state
    inside = np.zeros(w.N, bool)
    for k in range(w.M):
        inside |= np.all((w.lohi[k, 0] <= X) & (X <= w.lohi[k, 1]), axis=1)
    assert not inside.any()
    rc, W, _ = orc.sample_free(2024, 3000, 6, w.init, w.lohi, w.ss_lo, w.ss_hi, mp._lib.GOAL_BALL, w.goal_params(), goal_ct=0)
    assert np.array_equal(X[:3000], W)                                     # same stream, same order
    assert abs(X[1:-5].mean() - 0.5) < 0.01
    c.close()


# ---- graph persistence ---------------------------------------------------------------------------------

def test_graph_export_import_replan(orc, tmp_path):
    """Export the r-disc graph, load it into a fresh context (as another process would from disk), re-plan against a
    different obstacle set without the pair phase: masks, tree, costs identical to a from-scratch plan."""
    w = mp.workloads.make("t", 4000, 3, 60, 0.05, 0.15, seed=21, goal_radius=0.15)
    a = mp.Context(0)
    a.upload_samples(w.X); a.upload_boxes(w.lohi, w.ss_lo, w.ss_hi)
    colptr, rowval, nzval = a.rdisc_graph(w.r)
    np.savez(tmp_path / "graph.npz", colptr=colptr, rowval=rowval, nzval=nzval, r=w.r)
    ref = a.fmtstar(w.r, mp._lib.GOAL_BALL, w.goal_params())
    z = np.load(tmp_path / "graph.npz")
    b = mp.Context(0)
    b.upload_samples(w.X); b.upload_boxes(w.lohi, w.ss_lo, w.ss_hi)
    b.graph_import(float(z["r"]), z["colptr"], z["rowval"], z["nzval"])
    assert np.array_equal(b.graph_edges_free(), a.graph_edges_free())
    b.timing_reset()
    got = b.fmtstar(w.r, mp._lib.GOAL_BALL, w.goal_params())
    assert b.timing("rdisc_count")[1] == 0                       # the pair kernels never ran in the second context
    for k in ("status", "cost", "z", "collision_checks"):
        assert got[k] == ref[k]
    assert np.array_equal(got["A"], ref["A"]) and np.array_equal(got["C"], ref["C"]) and np.array_equal(got["path"], ref["path"])
    # new obstacles, same samples: only the sweep is redone, and it matches the oracle on the imported graph
    rng = np.random.default_rng(4)
    lohi2 = mp.workloads.make_boxes(rng, 30, 3, 0.05, 0.2, [w.init, w.goal_center])
    b.upload_boxes(lohi2, w.ss_lo, w.ss_hi)
    got2 = b.fmtstar(w.r, mp._lib.GOAL_BALL, w.goal_params())
    assert b.timing("rdisc_count")[1] == 0
    want2 = orc.fmtstar(w.X, w.r, orc.GOAL_BALL, w.goal_params(), lohi2, w.ss_lo, w.ss_hi)
    assert got2["status"] == want2["status"] and got2["cost"] == want2["cost"] and got2["collision_checks"] == want2["collision_checks"]
    assert np.array_equal(got2["A"] - 1, want2["A"])
    # malformed input is refused
    bad = z["rowval"].copy(); bad[[0, 1]] = bad[[1, 0]]
    with pytest.raises(mp.MPFMTError) as e:
        b.graph_import(w.r, z["colptr"], bad, z["nzval"])
    assert e.value.code == mp._lib.ERR_ARG
    a.close(); b.close()


# ---- 2-D SAT world -------------------------------------------------------------------------------------

def _shapes(fx_shapes):
    return [("circle", tuple(s[1]), s[2]) if s[0] == "circle" else ("polygon", [tuple(p) for p in s[1]]) for s in fx_shapes]


@pytest.mark.parametrize("name", ["ISRR_2H", "TRI_BALLS", "ISRR_POLY", "ISRR_POLY_WITH_SPIKE", "EMPTY_2D"])
def test_sat2d_goldens(ctx, name):
    fx = json.load(open(os.path.join(G, "shapes_2d.json")))
    z = np.load(os.path.join(G, "segments2d_%s.npz" % name))
    P, Q, n = z["P"], z["Q"], len(z["P"])
    ctx.upload_shapes2d(_shapes(fx["worlds"][name]))
    assert np.array_equal(mp._lib.unpack_bits(ctx.motions_free(P, Q), n), z["free_motion"])
    assert np.array_equal(mp._lib.unpack_bits(ctx.states_free(P), n), z["free_state"])
    ctx.upload_shapes2d(_shapes(fx["worlds"][name]), z["ss_lo"], z["ss_hi"])
    assert np.array_equal(mp._lib.unpack_bits(ctx.motions_free(P, Q), n), z["free_motion_ss"])
    assert np.array_equal(mp._lib.unpack_bits(ctx.states_free(P), n), z["free_state_ss"])
    for c in fx["known"]:
        if c[0] == name:
            ctx.upload_shapes2d(_shapes(fx["worlds"][name]))
            assert bool(mp._lib.unpack_bits(ctx.motions_free(np.array([c[1]]), np.array([c[2]])), 1)[0]) == c[3]
            assert bool(mp._lib.unpack_bits(ctx.states_free(np.array([c[1]])), 1)[0]) == c[4]


def test_sat2d_random_world_graph_and_plan(ctx, orc):
    """Random circles and convex polygons: point / edge / whole-graph masks against the oracle, then a full FMT* plan in
    the SAT world (same tree, cost and collision count as the oracle recursion fed with the oracle's masks)."""
    rng = np.random.default_rng(31)
    shapes = []
    for _ in range(14):
        c = 0.15 + 0.7 * rng.random(2)
        if rng.random() < 0.4:
            shapes.append(("circle", tuple(c), 0.03 + 0.05 * rng.random()))
        else:
            k = int(rng.integers(3, 9))
            ang = np.sort(rng.random(k) * 2 * np.pi)
            if np.max(np.diff(np.concatenate([ang, [ang[0] + 2 * np.pi]]))) > 0.95 * np.pi:
                continue                                                     # keep the hull well conditioned
            rad = 0.04 + 0.05 * rng.random()
            shapes.append(("polygon", [(c[0] + rad * np.cos(a), c[1] + rad * np.sin(a)) for a in ang]))
    init, goal = np.array([0.03, 0.03]), np.array([0.97, 0.97, 0.04])
    S = orc.Shapes2D(shapes)
    lo, hi = np.zeros(2), np.ones(2)
    N = 6000
    X = rng.random((N, 2)); X[0] = init; X[-1] = goal[:2]
    ctx.upload_samples(X)
    ctx.upload_shapes2d(shapes, lo, hi)
    r = 0.035
    colptr, rowval, nzval = ctx.rdisc_graph(r)
    c0, r0 = to0(colptr, rowval)
    Fo = orc.points_free_2d(X, S, lo, hi)
    assert np.array_equal(ctx.points_free(), Fo)
    eo = orc.graph_edges_free_2d(X, c0, r0, S, lo, hi)
    assert np.array_equal(ctx.graph_edges_free(), eo)
    cols = np.repeat(np.arange(1, N + 1), np.diff(colptr))
    assert np.array_equal(ctx.edges_free(rowval, cols), eo)
    assert 0.5 < mp._lib.unpack_bits(eo, len(r0)).mean() < 0.99
    got = ctx.fmtstar(r, mp._lib.GOAL_BALL, goal)
    want = orc.fmtstar_graph(X, c0, r0, nzval, eo, Fo, orc.GOAL_BALL, goal, np.zeros((0, 2, 2)), lo, hi, init_idx=0)
    assert got["status"] == want["status"] == 1
    assert got["cost"] == want["cost"] and got["collision_checks"] == want["collision_checks"]
    assert np.array_equal(got["A"] - 1, want["A"]) and np.array_equal(got["path"] - 1, want["path"])
    # sampler and checker switching
    Xs, _ = ctx.sample_free(5, 2000, init=init, goal_kind=mp._lib.GOAL_BALL, goal_params=goal, goal_ct=2)
    assert orc.unpack(orc.points_free_2d(Xs, S, lo, hi), 2000).all()
    ctx.upload_boxes(np.zeros((0, 2, 2)), lo, hi)                            # back to the (empty) AABB checker
    assert mp._lib.unpack_bits(ctx.points_free(), 2000).all()
    with pytest.raises(mp.MPFMTError) as e:
        ctx.upload_shapes2d([("polygon", [(0, 0), (1, 0), (0.2, 0.2), (0, 1)])])
    assert e.value.code == mp._lib.ERR_ARG


# ---- Dubins car ----------------------------------------------------------------------------------------

def _car_world(rng, N, M):
    X = np.column_stack([rng.random(N), rng.random(N), rng.random(N) * 2 * np.pi])
    c = 0.15 + 0.7 * rng.random((M, 2)); h = 0.02 + 0.05 * rng.random((M, 2))
    lohi = np.stack([c - h, c + h], axis=1)
    lo, hi = np.array([0.0, 0.0, 0.0]), np.array([1.0, 1.0, 2 * np.pi])
    return X, lohi, lo, hi


# Two comparisons per test, kept apart on purpose:
#   (1) PARITY against the oracle proper, whose sin / cos / atan2 / acos are the C library's -- an implementation the device shares
#       nothing with: costs to 1e-12 relative, memberships / masks / segment counts exact (no cost of these seeded sets lies within
#       1e-12 of the radius; the comparison reports one that does), the same word chosen, plans equal in status / cost / path, parents
#       equal except where two words tie to rounding;
#   (2) SELF-CONSISTENCY inside `with orc.device_math():` -- the oracle's second build compiles the product's own mp_math.h, so every
#       tie breaks identically and everything is compared bit for bit.  That block proves the pipeline around the transcendental
#       functions (word selection, ordering, sweep, recursion), not the functions themselves; tests/test_oracle.py checks those against
#       200-bit mpmath.

def _assert_car_graph_close(N, r, got, want):
    """(colptr0, rowval0, nzval) of the device against the libm oracle's: memberships equal except where the oracle's cost is within
    1e-12 r of the radius, costs of the common entries to 1e-12.  Returns the common entries' positions in both."""
    gk = np.repeat(np.arange(N), np.diff(got[0])) * N + got[1]
    wk = np.repeat(np.arange(N), np.diff(want[0])) * N + want[1]
    com, gi, wi = np.intersect1d(gk, wk, assume_unique=True, return_indices=True)
    og = np.setdiff1d(np.arange(len(gk)), gi); ow = np.setdiff1d(np.arange(len(wk)), wi)
    assert (np.abs(got[2][og] - r) <= 1e-12 * r).all() and (np.abs(want[2][ow] - r) <= 1e-12 * r).all()
    assert len(og) + len(ow) <= 1e-4 * len(com)
    assert np.allclose(got[2][gi], want[2][wi], rtol=1e-12, atol=0)
    return gi, wi


def _assert_car_plan_close(got, want, N):
    assert got["status"] == want["status"]
    assert abs(got["cost"] - want["cost"]) <= 1e-12 * want["cost"]
    fin = np.isfinite(want["C"])
    assert np.array_equal(np.isfinite(got["C"]), fin) and np.allclose(got["C"][fin], want["C"][fin], rtol=1e-12, atol=0)
    assert np.array_equal(got["path"] - 1, want["path"])
    # (two parents reached by words of equal length up to rounding tie differently under different sin / cos: a handful per plan)
    assert (got["A"] - 1 != want["A"]).sum() <= 0.005 * N
    assert abs(got["collision_checks"] - want["collision_checks"]) <= 0.005 * want["collision_checks"]


def test_dubins_steer_batch(ctx, orc):
    rng = np.random.default_rng(61)
    X0, _, _, _ = _car_world(rng, 4000, 1); X1, _, _, _ = _car_world(rng, 4000, 1)
    for rt in (0.05, 0.3, 2.0):
        cost, ctrl = ctx.dubins_steer(X0, X1, rt, 1.0)
        # (1) parity: the oracle with the C library's transcendental functions
        want = [orc.dubins(a, b, rt, 1.0) for a, b in zip(X0, X1)]
        wc = np.array([w[0] for w in want]); wu = np.array([w[1] for w in want])
        assert np.allclose(cost, wc, rtol=1e-12, atol=0)
        assert np.array_equal(ctrl[:, :, 1:], wu[:, :, 1:]) and np.allclose(ctrl, wu, rtol=0, atol=1e-12)      # same word, same durations
        # (2) self-consistency: ONE header on both sides (mp_math.h) -- costs, the winning word and its durations bit-identical
        with orc.device_math():
            want = [orc.dubins(a, b, rt, 1.0) for a, b in zip(X0, X1)]
        assert np.array_equal(cost, np.array([w[0] for w in want])) and np.array_equal(ctrl, np.array([w[1] for w in want]))


@pytest.mark.parametrize("N,rt,r", [(1500, 0.05, 0.25), (2500, 0.15, 0.3)])
def test_dubins_graph_sweep_and_plan(ctx, orc, N, rt, r):
    """Dubins backward sets, edge validity with the reference's arc waypoints, and a full plan, against the oracle: (1) the libm
    oracle -- memberships, masks, per-edge segment counts exact, costs to 1e-12, plan equal in status / cost / path; (2) the
    device-math build of the oracle: everything bit for bit, tree and collision_checks included."""
    rng = np.random.default_rng(70 + N)
    X, lohi, lo, hi = _car_world(rng, N, 12)
    X[0] = [0.05, 0.05, 0.6]; X[-1] = [0.95, 0.95, 0.8]
    ctx.upload_samples(X)
    ctx.upload_boxes(lohi, lo, hi, dw=2)
    colptr, rowval, nzval = ctx.dubins_graph(rt, 1.0, r)
    c0, r0 = to0(colptr, rowval)
    assert len(r0) > 5 * N                                                # a real graph, not a trivial one
    mask, nseg = ctx.dubins_graph_edges_free()
    assert 0.2 < mp._lib.unpack_bits(mask, len(r0)).mean() < 0.999
    goal = np.array([0.95, 0.95, 0.08])
    got = ctx.dubins_fmtstar(rt, 1.0, r, mp._lib.GOAL_BALL, goal)
    # (1) parity
    oc, orow, oval = orc.dubins_graph(X, rt, 1.0, r)
    gi, wi = _assert_car_graph_close(N, r, (c0, r0, nzval), (oc, orow, oval))
    omask, onseg = orc.dubins_graph_edges_free(X, rt, 1.0, oc, orow, lohi, lo, hi)
    assert np.array_equal(mp._lib.unpack_bits(mask, len(r0))[gi], orc.unpack(omask, len(orow))[wi]) and np.array_equal(nseg[gi], onseg[wi])
    _assert_car_plan_close(got, orc.dubins_fmtstar(X, rt, 1.0, oc, orow, oval, orc.GOAL_BALL, goal, lohi, lo, hi, init_idx=0), N)
    # (2) self-consistency (the oracle compiled with the device's mp_math.h)
    with orc.device_math():
        oc, orow, oval = orc.dubins_graph(X, rt, 1.0, r)
        omask, onseg = orc.dubins_graph_edges_free(X, rt, 1.0, oc, orow, lohi, lo, hi)
        want = orc.dubins_fmtstar(X, rt, 1.0, oc, orow, oval, orc.GOAL_BALL, goal, lohi, lo, hi, init_idx=0)
    assert np.array_equal(c0, oc) and np.array_equal(r0, orow) and np.array_equal(nzval, oval)
    assert np.array_equal(mask, omask) and np.array_equal(nseg, onseg)
    assert got["status"] == want["status"]
    assert np.array_equal(got["A"] - 1, want["A"]) and np.array_equal(got["path"] - 1, want["path"])
    assert got["collision_checks"] == want["collision_checks"]
    assert np.array_equal(got["C"], want["C"]) and got["cost"] == want["cost"]


# ---- Reeds-Shepp car -----------------------------------------------------------------------------------

def _rs_batch(want):
    wc = np.array([w[0] for w in want])
    wl = np.array([len(w[1]) for w in want])
    wu = np.zeros((len(want), 5, 3))
    for i, w in enumerate(want):
        wu[i, :len(w[1])] = w[1]
    return wc, wl, wu


def test_reedsshepp_steer_batch(ctx, orc):
    rng = np.random.default_rng(62)
    X0, _, _, _ = _car_world(rng, 4000, 1); X1, _, _, _ = _car_world(rng, 4000, 1)
    X1[:50] = X0[:50]                                                   # coincident poses
    X1[50:100, 2] = X0[50:100, 2]                                       # pure translations
    for rt in (0.05, 0.3, 2.0):
        cost, ctrl, nsegs = ctx.reedsshepp_steer(X0, X1, rt, 1.0)
        # (1) parity: libm oracle -- costs to 1e-12, the same word (segment count, directions, gears), durations to 1e-12
        wc, wl, wu = _rs_batch([orc.reedsshepp(a, b, rt, 1.0) for a, b in zip(X0, X1)])
        assert np.allclose(cost, wc, rtol=1e-12, atol=1e-15)
        assert np.array_equal(nsegs, wl) and np.array_equal(ctrl[:, :, 1:], wu[:, :, 1:]) and np.allclose(ctrl, wu, rtol=0, atol=1e-12)
        # (2) self-consistency: same transcendental functions on both sides (mp_math.h), so ties between words resolve identically
        with orc.device_math():
            wc, wl, wu = _rs_batch([orc.reedsshepp(a, b, rt, 1.0) for a, b in zip(X0, X1)])
        assert np.array_equal(cost, wc)
        assert np.array_equal(nsegs, wl) and np.array_equal(ctrl, wu)
        for i in range(len(wc)):                                        # segments past nsegs are zero
            assert not ctrl[i, nsegs[i]:].any()


@pytest.mark.parametrize("N,rt,r", [(1500, 0.05, 0.2), (2500, 0.15, 0.25)])
def test_reedsshepp_graph_sweep_and_plan(ctx, orc, N, rt, r):
    """Reeds-Shepp inball sets (column v holds d(v, w)), edge validity over the reference's waypoints, and a full plan
    through the symmetric recursion: (1) against the libm oracle -- memberships, masks, segment counts exact, costs to 1e-12, plan equal
    in status / cost / path, parents equal up to a handful of ties; (2) against the device-math build of the oracle: bit for bit."""
    rng = np.random.default_rng(90 + N)
    X, lohi, lo, hi = _car_world(rng, N, 12)
    X[0] = [0.05, 0.05, 0.6]; X[-1] = [0.95, 0.95, 0.8]
    ctx.upload_samples(X)
    ctx.upload_boxes(lohi, lo, hi, dw=2)
    colptr, rowval, nzval = ctx.reedsshepp_graph(rt, 1.0, r)
    c0, r0 = to0(colptr, rowval)
    assert len(r0) > 5 * N
    mask, nseg = ctx.reedsshepp_graph_edges_free()
    assert 0.2 < mp._lib.unpack_bits(mask, len(r0)).mean() < 0.999
    goal = np.array([0.95, 0.95, 0.08])
    got = ctx.reedsshepp_fmtstar(rt, 1.0, r, mp._lib.GOAL_BALL, goal)
    # (1) parity
    oc, orow, oval = orc.rs_graph(X, rt, 1.0, r)
    gi, wi = _assert_car_graph_close(N, r, (c0, r0, nzval), (oc, orow, oval))
    omask, onseg = orc.car_graph_edges_free(2, X, rt, 1.0, oc, orow, lohi, lo, hi)
    assert np.array_equal(mp._lib.unpack_bits(mask, len(r0))[gi], orc.unpack(omask, len(orow))[wi]) and np.array_equal(nseg[gi], onseg[wi])
    want = orc.rs_fmtstar(X, rt, 1.0, oc, orow, oval, orc.GOAL_BALL, goal, lohi, lo, hi, init_idx=0)
    assert want["status"] == 1
    _assert_car_plan_close(got, want, N)
    # (2) self-consistency.  (Reeds-Shepp words of the C|C|C kind have length = turning radius x heading change, so two parents reached
    # by such words tie up to rounding; with one set of transcendental functions on both sides the ties break identically.)
    with orc.device_math():
        oc, orow, oval = orc.rs_graph(X, rt, 1.0, r)
        omask, onseg = orc.car_graph_edges_free(2, X, rt, 1.0, oc, orow, lohi, lo, hi)
        want = orc.rs_fmtstar(X, rt, 1.0, oc, orow, oval, orc.GOAL_BALL, goal, lohi, lo, hi, init_idx=0)
    assert np.array_equal(c0, oc) and np.array_equal(r0, orow) and np.array_equal(nzval, oval)
    assert np.array_equal(mask, omask) and np.array_equal(nseg, onseg)
    assert got["status"] == want["status"] == 1
    assert np.array_equal(got["A"] - 1, want["A"]) and np.array_equal(got["path"] - 1, want["path"])
    assert got["collision_checks"] == want["collision_checks"]
    assert np.array_equal(got["C"], want["C"]) and got["cost"] == want["cost"]
    # a dubins graph on the same context afterwards replaces the Reeds-Shepp one (and the other way round)
    ctx.dubins_graph(rt, 1.0, r)
    with pytest.raises(Exception):
        ctx.reedsshepp_graph_edges_free()


# ---- Monte-Carlo collision probability of edges (BASELINE configs[4]) -------------------------------------------------

@pytest.mark.parametrize("d,M,sigma,R", [(2, 20, 0.02, 3000), (6, 200, 0.03, 1500), (3, 300, 0.05, 1000), (6, 0, 0.1, 500)])
def test_mc_edges_match_scalar_loop(ctx, orc, d, M, sigma, R):
    """Per-edge colliding-rollout counts equal the scalar loop's exactly (integer noise sums, unfused fp64)."""
    rng = np.random.default_rng(800 + d + M)
    X, lohi = random_world(rng, 400, d, M, 0.04, 0.12)
    lo, hi = np.full(d, 0.02), np.full(d, 0.98)
    ctx.upload_samples(X); ctx.upload_boxes(lohi, lo, hi)
    src = rng.integers(1, 401, 60); dst = rng.integers(1, 401, 60)
    got = ctx.mc_edges_collision(src, dst, sigma, R, seed=77)
    want = orc.mc_edges(X, src - 1, dst - 1, sigma, R, 77, lohi, lo, hi)
    assert np.array_equal(got, want)
    assert got.min() >= 0 and got.max() <= R
    if M > 0:
        assert 0 < (got > 0).sum()                       # some edges do hit something
    # sigma = 0 reproduces the deterministic check
    det = ctx.mc_edges_collision(src, dst, 0.0, 7, seed=1)
    free = mp._lib.unpack_bits(ctx.edges_free(src, dst), len(src))
    assert np.array_equal(det, np.where(free, 0, 7))


def test_mc_one_edge_many_rollouts(ctx, orc):
    """1e6 rollouts of one edge (the configuration BASELINE.json names): the estimate is stable across seeds to the
    binomial error and a 20k-rollout prefix equals the scalar loop."""
    X = np.array([[0.2, 0.2], [0.8, 0.2]])
    lohi = np.array([[[0.45, 0.26], [0.55, 0.5]]])                 # a box 2 sigma above the segment
    ctx.upload_samples(X); ctx.upload_boxes(lohi, np.zeros(2), np.ones(2))
    p = [ctx.mc_edges_collision([1], [2], 0.03, 1_000_000, seed=s)[0] / 1e6 for s in (1, 2, 3)]
    assert 1e-3 < p[0] < 1e-2 and max(p) - min(p) < 6 * np.sqrt(p[0] / 1e6)
    assert ctx.mc_edges_collision([1], [2], 0.03, 20000, seed=1)[0] == orc.mc_edges(X, [0], [1], 0.03, 20000, 1, lohi, np.zeros(2), np.ones(2))[0]


@pytest.mark.parametrize("d,M,sigma,R", [(2, 20, 0.02, 3000), (6, 200, 0.03, 1500), (3, 150, 0.05, 1000)])
def test_mc_importance_sampling_matches_scalar_loop(ctx, orc, d, M, sigma, R):
    """The importance-sampling estimator (mixture of the nominal noise and the noise shifted towards the closest obstacle point,
    Irwin-Hall likelihood-ratio weights quantised to 2^-40): the per-edge integer weight sums equal the scalar loop's exactly."""
    rng = np.random.default_rng(900 + d + M)
    X, lohi = random_world(rng, 400, d, M, 0.04, 0.12)
    lo, hi = np.full(d, 0.02), np.full(d, 0.98)
    ctx.upload_samples(X); ctx.upload_boxes(lohi, lo, hi)
    src = rng.integers(1, 401, 40); dst = rng.integers(1, 401, 40)
    p, raw = ctx.mc_edges_collision_is(src, dst, sigma, R, seed=78)
    want = orc.mc_is_edges(X, src - 1, dst - 1, sigma, R, 78, lohi, lo, hi)
    assert np.array_equal(raw, want)
    assert p.min() >= 0.0 and p.max() <= 2.0 and (raw > 0).sum() > 0


@pytest.mark.parametrize("d,M,sigma,R", [(2, 20, 0.02, 3000), (6, 200, 0.03, 1500), (3, 150, 0.05, 1000), (8, 60, 0.04, 600)])
def test_adaptive_importance_sampling_matches_scalar_loop(ctx, orc, d, M, sigma, R):
    """The ADAPTIVE estimator (pilot with inflated noise -> likelihood-ratio-weighted mean of the colliding perturbations = the shift of
    the mixture's second component): the per-edge shifts and integer weight sums equal the scalar loop's exactly, edges without a
    collision in the pilot fall back to plain Monte Carlo."""
    rng = np.random.default_rng(950 + d + M)
    X, lohi = random_world(rng, 400, d, M, 0.04, 0.12)
    lo, hi = np.full(d, 0.02), np.full(d, 0.98)
    ctx.upload_samples(X); ctx.upload_boxes(lohi, lo, hi)
    src = rng.integers(1, 401, 40); dst = rng.integers(1, 401, 40)
    p, raw, sh = ctx.mc_edges_collision_ais(src, dst, sigma, R, seed=79)
    want, wsh = orc.mc_ais_edges(X, src - 1, dst - 1, sigma, R, 79, lohi, lo, hi)
    assert np.array_equal(sh, wsh)
    assert np.array_equal(raw, want)
    assert np.abs(sh).max() <= 3.0 and (np.abs(sh).max(axis=1) > 0).sum() > 0
    plain = ctx.mc_edges_collision(src, dst, sigma, R, seed=79)
    none = np.abs(sh).max(axis=1) == 0                       # no collision in the pilot: every weight is 1 = the plain hit count
    assert np.array_equal(raw[none], plain[none].astype(np.uint64) << np.uint64(40))


def test_adaptive_importance_sampling_in_r6_among_200_boxes(ctx):
    """BASELINE configs[4] in the world it names (R^6, 200 boxes): over graph edges with collision probability between 1e-4 and 1e-2 the
    adaptive estimator is unbiased against long plain runs and its variance at equal rollouts is several times smaller (median ratio
    measured ~8; asserted >= 3; profiles/r04_mc_adaptive_is.txt has the study)."""
    w = mp.workloads.north_star(20000)
    ctx.upload_samples(w.X); ctx.upload_boxes(w.lohi, w.ss_lo, w.ss_hi)
    colptr, rowval, _ = ctx.rdisc_graph(w.r * 1.6)
    free = mp._lib.unpack_bits(ctx.graph_edges_free(), len(rowval))
    cols = np.repeat(np.arange(1, w.N + 1), np.diff(colptr))
    rng = np.random.default_rng(6)
    pick = rng.choice(np.flatnonzero(free), 192, replace=False)
    src, dst = rowval[pick].astype(np.int64), cols[pick].astype(np.int64)
    sigma, n, S = 0.02, 50_000, 16
    ref = np.mean([ctx.mc_edges_collision(src, dst, sigma, 1_000_000, seed=500 + s) / 1e6 for s in range(4)], axis=0)
    sel = (ref >= 1e-4) & (ref < 1e-2)
    assert sel.sum() >= 8, int(sel.sum())
    mc = np.array([ctx.mc_edges_collision(src[sel], dst[sel], sigma, n, seed=s) / n for s in range(S)])
    ais = np.array([ctx.mc_edges_collision_ais(src[sel], dst[sel], sigma, n, seed=s)[0] for s in range(S)])
    vr = mc.var(axis=0, ddof=1) / np.maximum(ais.var(axis=0, ddof=1), 1e-300)
    assert np.median(vr) >= 3.0, (np.median(vr), np.percentile(vr, [25, 75]))
    z = np.abs(ais.mean(axis=0) - ref[sel]) / np.sqrt(ais.var(axis=0, ddof=1) / S + ref[sel] / 4e6)
    assert z.max() < 6.0, z.max()


def test_mc_importance_sampling_reduces_the_variance(ctx):
    """A rare collision (p ~ 5e-5 at 20 000 rollouts: plain Monte Carlo sees 0, 1 or 2 hits): over 40 seeds both estimators agree in
    the mean and the importance-sampling one has at least ten times less variance (measured: ~40 times)."""
    X = np.array([[0.2, 0.2], [0.8, 0.25]])
    lohi = np.array([[[0.45, 0.36], [0.6, 0.6]], [[0.1, 0.7], [0.3, 0.9]]])
    ctx.upload_samples(X); ctx.upload_boxes(lohi, np.zeros(2), np.ones(2))
    n, sigma = 20000, 0.045
    mc = np.array([ctx.mc_edges_collision([1], [2], sigma, n, seed=s)[0] / n for s in range(40)])
    isv = np.array([ctx.mc_edges_collision_is([1], [2], sigma, n, seed=s)[0][0] for s in range(40)])
    se = np.sqrt(mc.var(ddof=1) / 40 + isv.var(ddof=1) / 40)
    assert abs(mc.mean() - isv.mean()) < 4 * se, (mc.mean(), isv.mean(), se)
    assert isv.var(ddof=1) * 10 < mc.var(ddof=1), (mc.var(ddof=1), isv.var(ddof=1))
    # a long plain run confirms the level: 4e6 rollouts
    ref = sum(int(ctx.mc_edges_collision([1], [2], sigma, 1_000_000, seed=100 + s)[0]) for s in range(4)) / 4e6
    assert abs(ref - isv.mean()) < 5 * np.sqrt(ref / 4e6 + isv.var(ddof=1) / 40), (ref, isv.mean())


def test_mc_one_edge_many_rollouts_in_r6_among_200_boxes(ctx, orc):
    """BASELINE configs[4] at its own size on the north-star world: 1e6 rollouts of one graph edge in R^6 among the 200 AABBs.  Two
    edges of the workload's graph (one the deterministic check finds free, one it finds blocked); the estimate is stable across
    seeds to the binomial error, a 20k-rollout prefix equals the scalar loop, and sigma = 0 reproduces the deterministic answer."""
    w = mp.workloads.north_star(20000)
    ctx.upload_samples(w.X); ctx.upload_boxes(w.lohi, w.ss_lo, w.ss_hi)
    colptr, rowval, _ = ctx.rdisc_graph(w.r * 1.6)
    free = mp._lib.unpack_bits(ctx.graph_edges_free(), len(rowval))
    cols = np.repeat(np.arange(1, w.N + 1), np.diff(colptr))
    picks = [int(np.flatnonzero(free)[len(rowval) // 7]), int(np.flatnonzero(~free)[5])]
    sigma = 0.02
    for e in picks:
        src, dst = [int(rowval[e])], [int(cols[e])]
        p = [ctx.mc_edges_collision(src, dst, sigma, 1_000_000, seed=s)[0] / 1e6 for s in (11, 12, 13)]
        assert max(p) - min(p) <= 6 * np.sqrt(max(p[0] * (1 - p[0]), 1e-6) / 1e6) + 1e-6, p
        pre = ctx.mc_edges_collision(src, dst, sigma, 20000, seed=11)[0]
        assert pre == orc.mc_edges(w.X, [src[0] - 1], [dst[0] - 1], sigma, 20000, 11, w.lohi, w.ss_lo, w.ss_hi)[0]
        det = ctx.mc_edges_collision(src, dst, 0.0, 5, seed=1)[0]
        assert det == (0 if free[e] else 5)
    assert p[0] > 0.3                                            # the blocked edge collides in a large share of its perturbed copies


# ---- closest obstacle points in a Mahalanobis metric ---------------------------------------------------

def _spd(rng, d):
    A = rng.standard_normal((d, d))
    return (A @ A.T + 0.3 * np.eye(d)) * 10 ** rng.uniform(-1, 1)


@pytest.mark.parametrize("d,M,n", [(1, 3, 500), (2, 12, 4000), (3, 40, 3000), (6, 200, 2000), (8, 30, 1500), (12, 10, 600)])
def test_closest_boxes_match_oracle(ctx, orc, d, M, n):
    """closest(p, CC, W) / closeR(p, CC, W, r2) over boxes (boxesND.jl:61-86 through bvls.jl): same winning box, same
    (point, box) pairs on which bvls exhausts its iterations, d2 / closest points to rounding (the device solves the projected
    problems by Cholesky on the normal equations, the oracle by Householder QR)."""
    rng = np.random.default_rng(300 + d)
    _, lohi = random_world(rng, 1, d, M, 0.02, 0.15)
    P = rng.random((n, d)) * 1.3 - 0.15
    P[: n // 8] = (lohi[0, 0] + (lohi[0, 1] - lohi[0, 0]) * rng.random((n // 8, d)))     # inside box 1
    P[n // 8] = lohi[0, 0]                                                               # a corner
    W = _spd(rng, d)
    ctx.upload_samples(P[:2]); ctx.upload_boxes(lohi, None, None)
    d2, v, k, fails = ctx.closest(P, W)
    od2, ov, ok, obad = orc.closest_boxes(P, lohi, W)
    assert fails == obad
    same = (k - 1 == ok)
    assert same.mean() > 0.999                                           # two boxes at equal distance to rounding may swap
    scale = np.maximum(od2, 1e-6 * np.trace(W))
    assert np.all(np.abs(d2 - od2) <= 1e-9 * scale)
    assert np.abs(v[same] - ov[same]).max() <= 1e-8
    # closeR at a radius that keeps a handful of boxes per point
    r2 = float(np.quantile(od2[np.isfinite(od2)], 0.5)) * 4 + 1e-3
    ptr, idx, dd, vv, fails2 = ctx.closeR(P, W, r2)
    optr, oidx, odd, ovv = orc.closeR_boxes(P, lohi, W, r2)
    assert fails2 == obad
    if np.array_equal(ptr - 1, optr) and np.array_equal(idx - 1, oidx):
        assert np.allclose(dd, odd, rtol=1e-9, atol=1e-9 * np.trace(W) * 1e-6) and np.abs(vv - ovv).max() <= 1e-8
    else:                                                                # only entries within rounding of r2 or of each other may differ
        cnt_diff = np.abs(np.diff(ptr) - np.diff(optr)).sum()
        assert cnt_diff <= 2 + len(idx) // 1000
    for i in range(0, n, max(n // 50, 1)):                               # ascending lists
        assert (np.diff(dd[ptr[i] - 1:ptr[i + 1] - 1]) >= 0).all()
    assert len(idx) > n // 4                                             # a real workload, not empty lists
    # capacity protocol of the C ABI
    import ctypes as C
    from motionplanning_jl_amd._lib import _dp, _ip
    tot = C.c_int64(); pbuf = np.empty(n + 1, dtype=np.int64)
    rc = ctx._L.mpfmt_closeR(ctx._h, _dp(np.ascontiguousarray(P)), n, _dp(np.ascontiguousarray(W)), r2, _ip(pbuf), 1,
                             _ip(np.empty(1, dtype=np.int64)), _dp(np.empty(1)), _dp(np.empty(d)), C.byref(tot), None)
    assert rc == mp._lib.ERR_CAPACITY and tot.value == len(idx) and np.array_equal(pbuf, ptr)


def test_closest_boxes_edge_cases(ctx, orc):
    P = np.array([[0.5, 0.5], [0.1, 0.9]])
    ctx.upload_samples(P); ctx.upload_boxes(np.zeros((0, 2, 2)), None, None, dw=2)
    d2, v, k, fails = ctx.closest(P, np.eye(2))
    assert np.isinf(d2).all() and (k == 0).all() and np.array_equal(v, P) and fails == 0      # (Inf, p)   boxesND.jl:73
    ptr, idx, dd, vv, _ = ctx.closeR(P, np.eye(2), 1.0)
    assert np.array_equal(ptr, [1, 1, 1]) and len(idx) == 0
    lohi = np.array([[[0.2, 0.2], [0.4, 0.6]]])
    ctx.upload_boxes(lohi, None, None)
    d2, v, k, fails = ctx.closest(np.array([[0.9, 0.5], [0.0, 0.0], [0.3, 0.9]]), np.eye(2))   # Euclidean projections
    assert np.allclose(v, [[0.4, 0.5], [0.2, 0.2], [0.3, 0.6]], atol=1e-15) and np.allclose(d2, [0.25, 0.08, 0.09], atol=1e-15)
    d2, v, k, fails = ctx.closest(np.zeros((0, 2)), np.eye(2))
    assert len(d2) == 0
    with pytest.raises(mp.MPFMTError):
        ctx.closest(P, None)                                             # boxes have no unweighted method (boxesND.jl:61)
    with pytest.raises(mp.MPFMTError):
        ctx.closest(P, np.array([[1.0, 2.0], [2.0, 1.0]]))               # not positive definite: chol(W) throws
    with pytest.raises(mp.MPFMTError):
        ctx.closest(P, np.array([[1.0, 0.5], [0.0, 1.0]]))               # not symmetric


def test_closest_shapes_match_oracle(ctx, orc):
    """closest / closeR over circles, convex polygons and compounds (SAT2D.jl:208-285), Euclidean and weighted."""
    rng = np.random.default_rng(77)
    shapes = [("circle", (0.3, 0.4), 0.1), ("polygon", [(0.6, 0.1), (0.9, 0.2), (0.8, 0.5), (0.55, 0.4)]), ("circle", (0.7, 0.8), 0.15),
              ("polygon", [(0.1, 0.7), (0.3, 0.7), (0.3, 0.9), (0.1, 0.9)]), ("polygon", [(0.45, 0.6), (0.55, 0.55), (0.5, 0.75)])]
    S = orc.Shapes2D(shapes)
    P = rng.random((6000, 2)) * 1.2 - 0.1
    ctx.upload_samples(P[:2]); ctx.upload_shapes2d(shapes)
    try:
        for W in (None, _spd(rng, 2), np.diag([4.0, 0.25]), _spd(rng, 2)):
            d2, v, k, fails = ctx.closest(P, W)
            od2, ov, ok, obad = orc.closest_shapes(P, S, W)
            assert fails == obad
            same = (k - 1 == ok)
            assert same.mean() > 0.999
            assert np.abs(d2 - od2).max() <= 1e-7                        # circles stop at |f| <= 1e-8: last Newton step may differ
            assert np.abs(v[same] - ov[same]).max() <= 1e-6
            poly = same & np.isin(ok, [1, 3, 4])
            assert np.abs(d2[poly] - od2[poly]).max() <= 1e-12 and np.abs(v[poly] - ov[poly]).max() <= 1e-12
            if W is not None:
                r2 = 0.05 * np.trace(W)
                ptr, idx, dd, vv, _ = ctx.closeR(P, W, r2)
                optr, oidx, odd, ovv = orc.closeR_shapes(P, S, W, r2)
                assert np.abs(np.diff(ptr) - np.diff(optr)).sum() <= 3
                if np.array_equal(idx - 1, oidx):
                    assert np.abs(dd - odd).max() <= 1e-7 and np.abs(vv - ovv).max() <= 1e-6
                assert len(idx) > 1000
    finally:
        ctx.upload_boxes(np.zeros((0, 2, 2)), None, None, dw=2)          # back to the box checker for the tests that follow


# ---- the single-synchronisation step (mpfmt_graph_step_device) -------------------------------------------------------

def _resident_graph(ctx, N):
    import torch
    from motionplanning_jl_amd.distributed import DevArray
    cp, rv, nz, fr = ctx.graph_device_ptrs()
    torch.cuda.synchronize()                  # (graph_sweep_device only enqueues on the ctx's own stream; torch copies on another)
    colptr = torch.as_tensor(DevArray(cp, N + 1, "<i8"), device="cuda:0").cpu().numpy()
    nnz = int(colptr[-1])
    if nnz == 0:
        return colptr, np.zeros(0, np.int32), np.zeros(0), np.zeros(0, np.int64)
    rowval = torch.as_tensor(DevArray(rv, nnz, "<i4"), device="cuda:0").cpu().numpy()
    nzval = torch.as_tensor(DevArray(nz, nnz, "<f8"), device="cuda:0").cpu().numpy()
    free = torch.as_tensor(DevArray(fr, (nnz + 63) // 64, "<i8"), device="cuda:0").cpu().numpy()
    return colptr, rowval, nzval, free


@pytest.mark.parametrize("world", [1, 3])
def test_graph_step_device_equals_build_plus_sweep(orc, world):
    """mpfmt_graph_step_device (one host synchronisation; sizes of the previous identical step taken on trust, validated
    afterwards) gives the same resident graph and mask as graph_build_device + graph_sweep_device -- on the first call
    (careful path), on repeats (speculative path), when the samples change under the same (N, r) (speculation may or may
    not hold) and when they change so much that every trusted capacity is wrong (device flag voids the kernels, host
    redoes the step)."""
    rng = np.random.default_rng(4242)
    N, d, M = 20000, 4, 30
    X, lohi = random_world(rng, N, d, M, 0.03, 0.1)
    r = 0.11
    lo, hi = np.full(d, 0.01), np.full(d, 0.99)
    for g in range(world):
        a = mp.Context(0); b = mp.Context(0)
        for c in (a, b):
            c.set_shard(g, world); c.set_option("rebuild_index", 1)
            c.upload_samples(X); c.upload_boxes(lohi, lo, hi)
        Xs = [X, X, X, rng.random((N, d)), rng.random((N, d)),
              0.5 + 0.04 * rng.standard_normal((N, d)),             # a tight cluster: degrees explode, every capacity is too small
              rng.random((N, d))]                                   # and back: capacities far too large
        for it, Xi in enumerate(Xs):
            a.upload_samples(Xi); b.upload_samples(Xi)
            nnz_b = b.graph_build_device(r); b.graph_sweep_device()
            nnz_a = a.graph_step_device(r)
            assert nnz_a == nnz_b, (g, it)
            ga, gb = _resident_graph(a, N), _resident_graph(b, N)
            for u, v in zip(ga, gb):
                assert np.array_equal(u, v), (g, it)
        assert nnz_a > 10 * N // world // 4
