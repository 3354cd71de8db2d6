"""GPU parity tests (-m gpu) of the streaming r-disc mode (mpfmt_rdisc_stream): per-column reductions without the stored graph --
degree, best open parent (fmt.jl:73: first minimum of C[y] + d over the ascending neighbourhood), free-edge count -- against the same
reductions of the ORACLE's graph, and BASELINE configs[2] at the radius of fmt.jl:39 (R^12, N = 1e6, r = 0.625: ~4 700 neighbours per
sample in the interior, a CSC of ~10 GB that is never built) on sampled columns."""
import numpy as np
import pytest

import motionplanning_jl_amd as mp

pytestmark = pytest.mark.gpu
L = mp._lib


def _reduce_columns(oc, orow, oval, Cc, Hbits, free_bits=None):
    N = len(oc) - 1
    deg = np.diff(oc)
    parent = np.zeros(N, dtype=np.int64); cost = np.full(N, np.inf); fdeg = np.zeros(N, dtype=np.int64)
    for x in range(N):
        a, b = oc[x], oc[x + 1]
        rows = orow[a:b]
        if free_bits is not None:
            fdeg[x] = int(free_bits[a:b].sum())
        keep = Hbits[rows]
        if keep.any():
            c = Cc[rows[keep]] + oval[a:b][keep]
            k = int(np.argmin(c))                              # first minimum, rows ascending (findmin, fmt.jl:73)
            parent[x] = rows[keep][k] + 1; cost[x] = c[k]
    return deg, parent, cost, fdeg


@pytest.mark.parametrize("d,N,r,M", [(2, 5000, 0.05, 20), (3, 7001, 0.09, 40), (6, 12000, 0.45, 200), (12, 6000, 1.1, 30), (7, 64, 0.9, 3)])
def test_stream_reductions_equal_the_oracle_graph(orc, d, N, r, M):
    rng = np.random.default_rng(300 + d)
    X = rng.random((N, d))
    X[5] = X[6]                                                # a duplicate pair: equal costs from equal distances
    lohi = mp.workloads.make_boxes(rng, M, d, 0.05, 0.25, [])
    lo, hi = np.full(d, 0.02), np.full(d, 0.98)               # (some samples fall outside: the first-point test of statespaces.jl:155)
    Cc = np.round(rng.random(N) * 2.0, 2)                      # coarse costs: ties between parents do occur
    Hbits = rng.random(N) < 0.4
    H = L.pack_bits(Hbits)
    oc, orow, oval = orc.rdisc_graph(X, r)
    fb = orc.unpack(orc.graph_edges_free(X, oc, orow, lohi, lo, hi), len(orow))
    deg, parent, cost, fdeg = _reduce_columns(oc, orow, oval, Cc, Hbits, fb)
    with mp.Context(0) as c:
        c.upload_samples(X); c.upload_boxes(lohi, lo, hi)
        got = c.rdisc_stream(r, Cc, H, want_free=True)
        assert got["nnz"] == len(orow)
        assert np.array_equal(got["deg"], deg)
        assert np.array_equal(got["parent"], parent)
        assert np.array_equal(got["cost"], cost)
        assert np.array_equal(got["free_deg"], fdeg)
        # every sample open, no costs: the degrees alone; and the resident-graph entry points still work afterwards
        g2 = c.rdisc_stream(r)
        assert np.array_equal(g2["deg"], deg) and g2["nnz"] == len(orow)
        allopen = c.rdisc_stream(r, Cc)
        d2, p2, c2, _ = _reduce_columns(oc, orow, oval, Cc, np.ones(N, dtype=bool))
        assert np.array_equal(allopen["parent"], p2) and np.array_equal(allopen["cost"], c2)
        colptr, rowval, nzval = c.rdisc_graph(r)
        assert np.array_equal(colptr - 1, oc) and np.array_equal(rowval - 1, orow) and np.array_equal(nzval, oval)


def test_cfg3_at_the_fmt_radius_streams(orc):
    """BASELINE configs[2] at rm = 1 (src/planners/fmt.jl:39): R^12, N = 1e6, r = 0.625 -- degrees and best open parents of ALL columns in
    one pass; sampled columns against the oracle's KD-tree neighbourhoods."""
    w = mp.workloads.cfg3()
    N, d = w.N, w.d
    r = mp.workloads.fmt_radius(1.0, d, 1.0, N)
    assert abs(r - 0.625) < 0.01
    rng = np.random.default_rng(9)
    Cc = rng.random(N) * 3.0
    Hbits = rng.random(N) < 0.25
    with mp.Context(0) as c:
        c.upload_samples(w.X); c.upload_boxes(w.lohi, w.ss_lo, w.ss_hi)
        got = c.rdisc_stream(r, Cc, L.pack_bits(Hbits), want_free=True)
    deg = got["deg"]
    assert got["nnz"] == int(deg.sum()) and got["nnz"] % 2 == 0
    # (E[deg] = 4 716 for an INTERIOR sample; in the unit cube of R^12 nearly every sample is within r of several faces: the mean is ~770)
    assert 3.0e8 < got["nnz"] < 6.0e9, got["nnz"]
    kd = orc.KDTree(w.X)
    for v in rng.integers(0, N, size=60):
        oi, od = kd.inball(int(v), r)
        assert deg[v] == len(oi), v
        # free edges of the column: is_free_motion(V[y], V[v]) over its rows (statespaces.jl:153-158), against the oracle's edge test
        fb = orc.unpack(orc.edges_free(w.X, oi, np.full(len(oi), v, dtype=np.int64), w.lohi, w.ss_lo, w.ss_hi), len(oi))
        assert got["free_deg"][v] == int(fb.sum()), v
        keep = Hbits[oi]
        if keep.any():
            cst = Cc[oi[keep]] + od[keep]
            k = int(np.argmin(cst))
            assert got["parent"][v] == oi[keep][k] + 1 and got["cost"][v] == cst[k], v
        else:
            assert got["parent"][v] == 0 and np.isinf(got["cost"][v])
