"""GPU parity tests (-m gpu) of the path bench.py actually times (mpfmt_graph_step_device: speculative sizes, k_spec_check,
redo path) against the ORACLE, of the fp16 MFMA filter's no-false-negative claim on adversarial inputs, of the
out-of-bounds-parent counting rule, and of BASELINE.json configs[2] / configs[3] at full size.
"""
import numpy as np
import pytest

import motionplanning_jl_amd as mp
from test_gpu_parity import _resident_graph, random_world, graphs_both_paths, to0

pytestmark = pytest.mark.gpu
L = mp._lib


# ---- (a) the timed path against the oracle ---------------------------------------------------------------------------

@pytest.mark.parametrize("world,rounds", [(1, 1), (3, 1), (1, 0), (3, 0)])
def test_graph_step_device_against_oracle(orc, world, rounds):
    """Resident CSC + free mask of mpfmt_graph_step_device vs the oracle's graph and edge predicate: the first (careful)
    call, two speculative repeats, new samples under the same (N, r), a capacity-busting cluster (device flag voids the
    kernels, host redoes the step) and the way back (capacities far too large).  Both sweep kernels (round table / task headers)."""
    rng = np.random.default_rng(777)
    N, d, M = 12000, 4, 30
    X, lohi = random_world(rng, N, d, M, 0.03, 0.1)
    r = 0.12
    lo, hi = np.full(d, 0.01), np.full(d, 0.99)
    Xs = [X, X, X, rng.random((N, d)), 0.5 + 0.05 * rng.standard_normal((N, d)), rng.random((N, d))]
    refs = []
    for Xi in Xs:
        if refs and Xi is Xs[0]:
            refs.append(refs[0]); continue
        oc, orow, oval = orc.rdisc_graph(Xi, r)
        refs.append((oc, orow, oval, orc.graph_edges_free(Xi, oc, orow, lohi, lo, hi)))
    for g in range(world):
        with mp.Context(0) as c:
            c.set_shard(g, world); c.set_option("rebuild_index", 1)
            c.set_option("sweep_rounds", rounds)             # k_graph_sweep_rt (round table) or k_graph_sweep (task headers)
            c.upload_boxes(lohi, lo, hi)
            for it, (Xi, (oc, orow, oval, omask)) in enumerate(zip(Xs, refs)):
                c.upload_samples(Xi)
                nnz = c.graph_step_device(r)
                colptr, rowval, nzval, free = _resident_graph(c, N)
                deg = np.diff(colptr)
                own = np.flatnonzero(deg)
                if world == 1:
                    assert np.array_equal(colptr, oc) and np.array_equal(rowval, orow) and np.array_equal(nzval, oval), it
                    assert np.array_equal(free.view(np.uint64), omask), it
                else:
                    odeg = np.diff(oc)
                    assert np.array_equal(deg[own], odeg[own]), (g, it)
                    # entries of the owned columns, in CSC order, against the oracle's same columns
                    oidx = np.concatenate([np.arange(oc[v], oc[v + 1]) for v in own]) if len(own) else np.zeros(0, np.int64)
                    assert np.array_equal(rowval, orow[oidx]) and np.array_equal(nzval, oval[oidx]), (g, it)
                    assert np.array_equal(L.unpack_bits(free.view(np.uint64), nnz), orc.unpack(omask, len(orow))[oidx]), (g, it)
                assert nnz == deg.sum()


@pytest.mark.parametrize("world,blocks,halo,N", [(2, 1, 1, 9000), (8, 1, 1, 9000), (8, 0, 1, 9000), (8, 1, 0, 9000), (8, 0, 0, 9000),
                                                  (5, 1, 1, 14001), (8, 1, 1, 30011)])
def test_shards_partition_the_columns(orc, world, blocks, halo, N):
    """Every column belongs to exactly one shard: the per-rank degree arrays of a sharded step SUM to the oracle's degrees (a column no
    rank owns, or one that lost rows to a missing halo tile, shows here -- comparing only the columns a rank reports would not), under both
    cell-id orders (`shard_blocks`: block-major / row-major), with and without the shard + halo index (`index_halo`), with shard counts
    that leave holes in the block-major id range (odd cell counts along a cut axis), over a careful step, a speculative repeat and new
    samples; the owned columns' entries and edge bits against the oracle as well."""
    rng = np.random.default_rng(8800 + world * 10 + blocks * 2 + halo)
    d, M = 4, 24
    r = 0.12 if N < 20000 else 0.095                          # (cells per axis: 8 / 10 -- and 7 for the third set below: odd)
    X, lohi = random_world(rng, N, d, M, 0.03, 0.1)
    lo, hi = np.full(d, 0.0), np.full(d, 1.0)
    sets = [X, X, 0.02 + 0.96 * rng.random((N, d)) * np.array([1.0, 0.86, 1.0, 1.0])]
    refs = []
    for Xi in sets:
        if refs and Xi is sets[0]:
            refs.append(refs[0]); continue
        oc, orow, oval = orc.rdisc_graph(Xi, r)
        refs.append((oc, orow, oval, orc.unpack(orc.graph_edges_free(Xi, oc, orow, lohi, lo, hi), len(orow))))
    degsum = [np.zeros(N, np.int64) for _ in sets]
    for g in range(world):
        with mp.Context(0) as c:
            c.set_shard(g, world); c.set_option("rebuild_index", 1)
            c.set_option("shard_blocks", blocks); c.set_option("index_halo", halo)
            c.upload_boxes(lohi, lo, hi)
            for it, (Xi, (oc, orow, oval, obits)) in enumerate(zip(sets, refs)):
                c.upload_samples(Xi)
                nnz = c.graph_step_device(r)
                colptr, rowval, nzval, free = _resident_graph(c, N)
                deg = np.diff(colptr)
                assert nnz == deg.sum()
                degsum[it] += deg
                own = np.flatnonzero(deg)
                oidx = np.concatenate([np.arange(oc[v], oc[v + 1]) for v in own]) if len(own) else np.zeros(0, np.int64)
                assert np.array_equal(deg[own], np.diff(oc)[own]), (g, it)
                assert np.array_equal(rowval, orow[oidx]) and np.array_equal(nzval, oval[oidx]), (g, it)
                assert np.array_equal(L.unpack_bits(free.view(np.uint64), nnz), obits[oidx]), (g, it)
    for it, (oc, _, _, _) in enumerate(refs):
        assert np.array_equal(degsum[it], np.diff(oc)), it     # no column unowned, none owned twice, none short of a row


def test_sharded_step_after_another_graph_build_on_the_ctx(orc):
    """a sharded ctx skips the fill of its degree array when the last step's ordering pass left it zero (`deg_zero_valid`);
    a double-integrator build in between writes EVERY column's degree into the same array.  Sequence: sharded Euclidean step, DI graph
    (count + fill) on other samples, sharded step again -- graph, costs and mask of every step against the oracle, degrees summed
    over the ranks."""
    rng = np.random.default_rng(8899)
    N, d, M, r, world = 6000, 4, 20, 0.14, 3
    X, lohi = random_world(rng, N, d, M, 0.03, 0.1)
    lo, hi = np.full(d, 0.0), np.full(d, 1.0)
    oc, orow, oval = orc.rdisc_graph(X, r)
    obits = orc.unpack(orc.graph_edges_free(X, oc, orow, lohi, lo, hi), len(orow))
    Xdi = rng.random((1500, 4)) * np.array([1.0, 1.0, 0.5, 0.5])
    degsum = [np.zeros(N, np.int64) for _ in range(3)]
    for g in range(world):
        with mp.Context(0) as c:
            c.set_shard(g, world); c.set_option("rebuild_index", 1)
            c.upload_boxes(lohi, lo, hi)
            for it in range(3):
                if it == 2:
                    c.upload_samples(Xdi)
                    dc, drow, dval, dt = c.di_graph(1.0, 0.9)                # writes deg[] of all 1500 columns
                    assert dc[-1] - 1 == len(drow) and len(drow) > 0
                c.upload_samples(X)
                nnz = c.graph_step_device(r)
                colptr, rowval, nzval, free = _resident_graph(c, N)
                deg = np.diff(colptr)
                degsum[it] += deg
                own = np.flatnonzero(deg)
                oidx = np.concatenate([np.arange(oc[v], oc[v + 1]) for v in own])
                assert nnz == deg.sum() and np.array_equal(deg[own], np.diff(oc)[own]), (g, it)
                assert np.array_equal(rowval, orow[oidx]) and np.array_equal(nzval, oval[oidx]), (g, it)
                assert np.array_equal(L.unpack_bits(free.view(np.uint64), nnz), obits[oidx]), (g, it)
    for it in range(3):
        assert np.array_equal(degsum[it], np.diff(oc)), it


@pytest.mark.parametrize("d,N,r", [(2, 6000, 0.03), (3, 7001, 0.09), (6, 20000, 0.42)])
def test_half_build_equals_whole_build_and_the_oracle(orc, d, N, r):
    """The single-pass build in its half form (every pair tested once, the other column's record written to a foreign log) against
    the whole form and the oracle: graph, costs, free mask, over a careful first step, speculative repeats and new samples."""
    rng = np.random.default_rng(4000 + d)
    X, lohi = random_world(rng, N, d, 12, 0.05, 0.2)
    lo, hi = np.full(d, 0.0), np.full(d, 1.0)
    sets = [X, X, rng.random((N, d))]
    got = {}
    used = 0
    for half in (1, 0):
        with mp.Context(0) as c:
            c.set_option("rdisc_half", half); c.set_option("rebuild_index", 1)
            c.upload_boxes(lohi, lo, hi)
            for it, Xi in enumerate(sets):
                c.upload_samples(Xi)
                nnz = c.graph_step_device(r)
                assert half == 1 or c.stat("rdisc_half_used") == 0
                used = used + c.stat("rdisc_half_used")     # (new samples may outgrow a trusted capacity: that step is redone whole)
                got[(half, it)] = _resident_graph(c, N)
                assert nnz == got[(half, it)][0][-1]
    assert used >= 1                                          # the half form did run (a first estimate that overflows is redone whole)
    for it, Xi in enumerate(sets):
        a, b = got[(1, it)], got[(0, it)]
        for u, v in zip(a, b):
            assert np.array_equal(u, v), it
        if it != 1:
            oc, orow, oval = orc.rdisc_graph(Xi, r)
            assert np.array_equal(a[0], oc) and np.array_equal(a[1], orow) and np.array_equal(a[2], oval)
            assert np.array_equal(a[3].view(np.uint64), orc.graph_edges_free(Xi, oc, orow, lohi, lo, hi))


@pytest.mark.parametrize("d,N,r,dup", [(2, 50, 0.3, 0), (3, 64, 0.5, 7), (6, 65, 0.9, 0), (2, 200, 0.4, 40), (6, 1000, 0.7, 100), (4, 5000, 0.2, 0)])
def test_pairs_inside_one_tile_are_found_once(orc, d, N, r, dup):
    """The half build keeps the pairs of a tile's OWN chunk once (sign bits of (query, candidate <= query) masked off, the other column's
    record sent to the tile's own logs): worlds where most or all edges join two samples of one 64-sample tile -- a single tile, a tile
    and one sample, repeated samples (distance 0, distinct indices) -- under the fused edge tests, against the oracle; and the slice
    count of the pair kernel's work items is odd (kernels_rdisc_mfma.hip, mpfmt_slices_for)."""
    rng = np.random.default_rng(900 + N)
    X, lohi = random_world(rng, N, d, 9, 0.05, 0.25)
    if dup:
        X[rng.integers(0, N, size=dup)] = X[rng.integers(0, N, size=dup)]       # repeated samples
    lo, hi = np.full(d, 0.0), np.full(d, 1.0)
    oc, orow, oval = orc.rdisc_graph(X, r)
    omask = orc.graph_edges_free(X, oc, orow, lohi, lo, hi)
    with mp.Context(0) as c:
        c.set_option("rebuild_index", 1)
        c.upload_samples(X); c.upload_boxes(lohi, lo, hi)
        for it in range(3):
            nnz = c.graph_step_device(r)
            colptr, rowval, nzval, free = _resident_graph(c, N)
            assert nnz == len(orow) and np.array_equal(colptr, oc) and np.array_equal(rowval, orow) and np.array_equal(nzval, oval), it
            assert np.array_equal(free.view(np.uint64), omask), it
            if c.stat("rdisc_path_used") == 2:
                assert c.graph_stats()["slices"] % 2 == 1
        assert c.stat("rdisc_half_used") == 1 and c.stat("sweep_form") == 2


_ORACLE_GRAPHS, _ORACLE_MASKS = {}, {}


@pytest.mark.parametrize("form", [2, 1, 0])
@pytest.mark.parametrize("d,N,r,M", [(2, 6000, 0.03, 12), (3, 7001, 0.09, 40), (6, 20000, 0.42, 200),
                                     (7, 12000, 0.42, 200), (9, 10000, 0.55, 200), (12, 10000, 0.70, 200)])
def test_edge_tests_fused_into_the_half_build(orc, form, d, N, r, M):
    """The step's edge tests in their three forms -- 2: broad phase in the pair kernel's drain, flagged PAIRS tested before the logs are
    ordered, the ordering pass writes the mask; 1: flagged ENTRIES listed by the ordering pass and tested afterwards; 0: the whole
    sweep -- against the oracle's graph and mask, over a careful step, speculative repeats, new samples and new obstacles."""
    rng = np.random.default_rng(6000 + d)
    X, lohi = random_world(rng, N, d, M, 0.05, 0.25)
    lohi2 = mp.workloads.make_boxes(rng, max(M // 2, 1), d, 0.1, 0.3, [])
    lo, hi = np.full(d, 0.0), np.full(d, 1.0)
    seen = []
    with mp.Context(0) as c:
        c.set_option("fuse_broad", form); c.set_option("rebuild_index", 1)
        for Xi, boxes in ((X, lohi), (X, lohi), (rng.random((N, d)), lohi), (X, lohi2), (X, lohi2)):
            c.upload_samples(Xi); c.upload_boxes(boxes, lo, hi)
            nnz = c.graph_step_device(r)
            seen.append(c.stat("sweep_form"))
            colptr, rowval, nzval, free = _resident_graph(c, N)
            # (the oracle's graph of a sample set and its mask per obstacle set are the same for every form: computed once per world)
            kg = (d, N, r, Xi is X)
            if kg not in _ORACLE_GRAPHS: _ORACLE_GRAPHS[kg] = orc.rdisc_graph(Xi, r)
            oc, orow, oval = _ORACLE_GRAPHS[kg]
            km = kg + (boxes is lohi,)
            if km not in _ORACLE_MASKS: _ORACLE_MASKS[km] = orc.graph_edges_free(Xi, oc, orow, boxes, lo, hi)
            assert np.array_equal(colptr, oc) and np.array_equal(rowval, orow) and np.array_equal(nzval, oval)
            assert np.array_equal(free.view(np.uint64), _ORACLE_MASKS[km])
        # a sweep of the resident graph on its own, after the obstacles changed, is the whole sweep whatever the step did
        oc, orow, oval = _ORACLE_GRAPHS[(d, N, r, True)]
        c.upload_boxes(lohi, lo, hi)
        c.graph_sweep_device()
        free = _resident_graph(c, N)[3]
        assert np.array_equal(free.view(np.uint64), _ORACLE_MASKS[(d, N, r, True, True)])
    ok = {0, form} | ({1} if form == 2 else set())            # (0: not a single-pass build, or a trusted capacity did not hold; 1 under 2: not a half build)
    assert all(s in ok for s in seen), seen
    # (7 <= d <= 12: the drain's broad phase runs as chains of six axes and exists as form 2 only -- form 1's exact kernel is built for d <= 6)
    if d >= 6 and not (d > 6 and form == 1): assert form == 0 or form in seen, seen
    if d > 6 and form == 1: assert set(seen) == {0}, seen


@pytest.mark.parametrize("d,N,r,M", [(3, 7001, 0.09, 40), (6, 20000, 0.42, 200), (9, 10000, 0.55, 200)])
def test_step_with_and_without_the_side_stream(orc, d, N, r, M):
    """The step's two side-stream forks (sample masks beside the chunk lists; degree count and scan beside the flagged pairs' exact
    tests) against the same step with every kernel on one stream (option overlap = 0) and against the oracle: careful step, speculative
    repeats, new samples, on the ctx's own stream and on a caller's; and with the launch's last tiles cut into more slices than the rest."""
    import torch
    rng = np.random.default_rng(6000 + d)
    X, lohi = random_world(rng, N, d, M, 0.05, 0.25)
    lo, hi = np.full(d, 0.0), np.full(d, 1.0)
    X2 = rng.random((N, d))
    user = torch.cuda.Stream(device="cuda:0")
    got = {}
    for ov, st in ((1, None), (0, None), (1, user), (2, None)):
        with mp.Context(0) as c:
            c.set_option("overlap", 2 if ov else 0); c.set_option("rebuild_index", 1)      # (2: forced -- by default only from 65536 samples on)
            if ov == 2:           # the last third of the tiles cut into 7 slices instead of the launch's own (forced: these launches are small)
                c.set_option("mf_tail_min_items", 0); c.set_option("mf_tail_permille", 333); c.set_option("mf_tail_slices", 7)
            # (the ordering kernel's quarters: drawn from the counters whatever their length / by the default rule / a fixed share each)
            c.set_option("ord_draw", 2 if ov == 1 and st is None else (1 if ov == 2 else 0))
            if st is not None: c.set_stream(st.cuda_stream)
            outs = []
            for Xi in (X, X, X2, X2, X):
                c.upload_samples(Xi); c.upload_boxes(lohi, lo, hi)
                c.graph_step_device(r)
                assert c.stat("sweep_form") == 2 and c.stat("rdisc_half_used") == 1
                outs.append(_resident_graph(c, N))
            got[(ov, st is not None)] = outs
    for k in range(5):
        for key in ((0, False), (1, True), (2, False)):
            for u, v in zip(got[(1, False)][k], got[key][k]):
                assert np.array_equal(u, v)
    for k, Xi in ((0, X), (2, X2)):
        # (X and lohi are the fused test's first world -- same seed --, its oracle results are shared; X2 is this test's own)
        kg = (d, N, r, True) if Xi is X else None
        if kg is None: oc, orow, oval = orc.rdisc_graph(Xi, r)
        else:
            if kg not in _ORACLE_GRAPHS: _ORACLE_GRAPHS[kg] = orc.rdisc_graph(Xi, r)
            oc, orow, oval = _ORACLE_GRAPHS[kg]
        if kg is None: omask = orc.graph_edges_free(Xi, oc, orow, lohi, lo, hi)
        else:
            km = kg + (True,)
            if km not in _ORACLE_MASKS: _ORACLE_MASKS[km] = orc.graph_edges_free(Xi, oc, orow, lohi, lo, hi)
            omask = _ORACLE_MASKS[km]
        colptr, rowval, nzval, free = got[(1, False)][k]
        assert np.array_equal(colptr, oc) and np.array_equal(rowval, orow) and np.array_equal(nzval, oval)
        assert np.array_equal(free.view(np.uint64), omask)


@pytest.mark.parametrize("form", [2, 1])
def test_fused_edge_tests_among_many_overlapping_obstacles(orc, form):
    """256 large boxes (the last id is 255, the most the packed per-lane lists can name): most segments meet more than four of them,
    which sends them down the every-box paths of k_exact_pairs / k_sweep_pending."""
    rng = np.random.default_rng(4321)
    N, d, r, M = 9000, 3, 0.07, 256
    X = rng.random((N, d))
    lohi = mp.workloads.make_boxes(rng, M, d, 0.15, 0.35, [])
    lo, hi = np.full(d, 0.0), np.full(d, 1.0)
    oc, orow, oval = orc.rdisc_graph(X, r)
    want = orc.graph_edges_free(X, oc, orow, lohi, lo, hi)
    with mp.Context(0) as c:
        c.set_option("fuse_broad", form); c.set_option("rebuild_index", 1)
        c.upload_samples(X); c.upload_boxes(lohi, lo, hi)
        forms = []
        for it in range(3):
            c.graph_step_device(r)
            forms.append(c.stat("sweep_form"))
            colptr, rowval, nzval, free = _resident_graph(c, N)
            assert np.array_equal(rowval, orow) and np.array_equal(free.view(np.uint64), want), it
    assert form in forms, forms


@pytest.mark.parametrize("form", [2, 1])
def test_fused_edge_tests_survive_a_list_overflow(orc, form):
    """Option debug_small_lists shrinks the pending lists to 8 items: the kernels that read them return at once, the host finds the
    flag behind its synchronisation and sweeps the whole graph."""
    rng = np.random.default_rng(77)
    N, d, r = 9000, 3, 0.08
    X, lohi = random_world(rng, N, d, 30, 0.05, 0.25)
    lo, hi = np.full(d, 0.0), np.full(d, 1.0)
    oc, orow, oval = orc.rdisc_graph(X, r)
    want = orc.graph_edges_free(X, oc, orow, lohi, lo, hi)
    with mp.Context(0) as c:
        c.set_option("fuse_broad", form); c.set_option("debug_small_lists", 1); c.set_option("rebuild_index", 1)
        c.upload_samples(X); c.upload_boxes(lohi, lo, hi)
        for it in range(3):                                   # careful, then speculative
            c.graph_step_device(r)
            colptr, rowval, nzval, free = _resident_graph(c, N)
            assert np.array_equal(rowval, orow) and np.array_equal(free.view(np.uint64), want), it
            assert c.stat("sweep_form") == 0


def test_half_build_overflow_is_redone_whole(orc):
    """A cluster the capacity estimate does not expect: the half build's logs overflow, the count is redone whole (a half build
    has no fill pass to fall back to), and once a build has left its size hint the half form is tried again."""
    rng = np.random.default_rng(91)
    X = np.concatenate([0.5 + 0.004 * rng.standard_normal((1800, 3)), rng.random((900, 3))])      # (every pair of the cluster is an edge: columns of 1800+, under the ordering kernel's 2048)
    oc, orow, oval = orc.rdisc_graph(X, 0.2)
    with mp.Context(0) as c:
        c.upload_samples(X)
        seen = []
        for _ in range(3):
            colptr, rowval, nzval = c.rdisc_graph(0.2)
            assert np.array_equal(colptr - 1, oc) and np.array_equal(rowval - 1, orow) and np.array_equal(nzval, oval)
            seen.append((c.stat("pool_used"), c.stat("rdisc_half_used")))
        assert seen[0] == (0, 0)                              # estimate too small: two-pass build
        assert seen[1][0] == 1                                # sized by the first build's count


def test_trimmed_stress_in_suite(orc):
    """tools/stress.py with fixed seeds and a bounded budget: random sizes / dimensions / radii / obstacle counts / shards,
    both pair kernels, graph + costs + masks + the three-call graph_step_device sequence, all against the oracle."""
    import importlib.util
    import os
    p = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "stress.py")
    spec = importlib.util.spec_from_file_location("mpfmt_stress", p)
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    cases, edges = m.run(budget=40.0, seed=20260, maxd=12, max_cases=60)
    assert cases >= 20 and edges > 100000


# ---- (b) adversarial inputs for the fp16 filter ----------------------------------------------------------------------

def adversarial_set(rng, N, d, r, scale=1.0, offset=0.0):
    """Random cloud + pairs at distance exactly r, r(1 +- k ulp), r(1 +- 1e-12), r(1 +- 1e-9), duplicates, points hugging the
    low corner of the cloud (fp16-denormal normalised coordinates), axis-aligned and diagonal displacements."""
    X = rng.random((N, d))
    k = 0
    def put(p):
        nonlocal k
        X[k] = p; k += 1
    for _ in range(60):
        p = 0.2 + 0.6 * rng.random(d)
        for eps in (0.0, 2.3e-16, -2.3e-16, 1e-15, -1e-15, 1e-12, -1e-12, 1e-9, -1e-9, 3e-5, -3e-5):
            u = rng.standard_normal(d); u /= np.linalg.norm(u)
            put(p); put(p + r * (1 + eps) * u)
        ax = np.zeros(d); ax[rng.integers(0, d)] = 1.0
        put(p); put(p + r * ax); put(p - r * ax)
        put(p); put(p + r * np.ones(d) / np.sqrt(d))
        put(p); put(p.copy())                                          # duplicate
    for j in range(80):                                                 # the low corner: normalised coordinates 1e-7 .. 6e-5
        p = rng.random(d) * 10.0 ** rng.uniform(-7, -4.2)
        put(p)
        u = rng.random(d); u /= np.linalg.norm(u)
        put(p + r * (1 + (1e-12 if j & 1 else -1e-12)) * u)
    X[k] = 0.0; k += 1                                                  # the corner itself
    X[k] = 1.0; k += 1
    assert k < N
    return X * scale + offset


@pytest.mark.parametrize("d,r", [(1, 0.004), (2, 0.05), (3, 0.12), (6, 0.45), (7, 0.55), (12, 0.95)])
@pytest.mark.parametrize("scale,offset", [(1.0, 0.0), (1e-3, 0.0), (1e3, 0.0), (1.0, 1000.0)])
def test_filter_has_no_false_negatives_on_adversarial_pairs(orc, d, r, scale, offset):
    rng = np.random.default_rng(900 + d)
    N = 3500
    X = adversarial_set(rng, N, d, r, scale, offset)
    rr = r * scale
    with mp.Context(0) as c:
        c.upload_samples(X)
        (colptr, rowval, nzval), ran = graphs_both_paths(c, rr)
        assert 1 in ran
        oc, orow, oval = orc.rdisc_graph(X, rr)
        c0, r0 = to0(colptr, rowval)
        assert np.array_equal(c0, oc) and np.array_equal(r0, orow)
        assert np.array_equal(nzval, oval)
        if d <= 12 and d >= 2:
            assert 2 in ran, "the MFMA filter path did not run for d = %d" % d
        # the pairs sit on the threshold: the oracle must see members on both sides of it
        cols = np.repeat(np.arange(N), np.diff(oc))
        assert (oval >= rr * (1 - 1e-9)).sum() > 50


# ---- (e) out-of-bounds parents: in_state_space short circuit before the checker's count ------------------------------

def test_out_of_bounds_parent_counting(orc):
    from test_oracle import oob_parent_world
    X, lohi, lo, hi, r, goal = oob_parent_world()
    ref = orc.fmtstar(X, r, orc.GOAL_BALL, goal, lohi, lo, hi, checkpts=False, nn_mode=1)
    with mp.Context(0) as c:
        c.upload_samples(X); c.upload_boxes(lohi, lo, hi)
        seq = c.fmtstar(r, L.GOAL_BALL, goal, checkpts=False)
        for got in (seq, c.fmtstar_wavefront(r, L.GOAL_BALL, goal, single=True, checkpts=False),
                    c.fmtstar_wavefront(r, L.GOAL_BALL, goal, single=True, checkpts=False, eager=True)):
            assert got["status"] == ref["status"] and got["collision_checks"] == ref["collision_checks"]
            assert np.array_equal(got["A"] - 1, ref["A"]) and np.array_equal(got["C"], ref["C"])
            assert np.array_equal(got["path"] - 1, ref["path"])


# ---- (c) BASELINE.json configs[2] and configs[3] at full size ---------------------------------------------------------

def test_cfg3_r12_full_size_properties(orc):
    """PRM*-style all-pairs r-disc graph in R^12, N = 1e6 (BASELINE.json configs[2]): column contract, symmetry checksums,
    sampled columns against the oracle's KD-tree, sampled edge bits and the point mask against the oracle."""
    w = mp.workloads.cfg3()
    N = w.N
    with mp.Context(0) as c:
        c.upload_samples(w.X)
        c.upload_boxes(w.lohi, w.ss_lo, w.ss_hi)
        colptr, rowval, nzval = c.rdisc_graph(w.r)
        assert c.stat("rdisc_path_used") == 2
        nnz = len(rowval)
        deg = np.diff(colptr)
        assert colptr[0] == 1 and colptr[-1] == nnz + 1 and nnz % 2 == 0 and nnz > 2e7
        cols = np.repeat(np.arange(1, N + 1, dtype=np.int64), deg)
        assert not np.any(rowval == cols)
        dd = np.diff(rowval); same = cols[1:] == cols[:-1]
        assert np.all(dd[same] > 0)
        del dd, same
        assert nzval.max() <= w.r * (1 + 1e-12) and nzval.min() > 0
        k1 = cols * N + rowval; k2 = rowval * N + cols
        assert int(k1.sum()) == int(k2.sum()) and int(np.bitwise_xor.reduce(k1)) == int(np.bitwise_xor.reduce(k2))
        del k1, k2
        kd = orc.KDTree(w.X)
        rng = np.random.default_rng(5)
        for v in rng.integers(0, N, size=150):
            oi, od = kd.inball(int(v), w.r)
            a, b = colptr[v] - 1, colptr[v + 1] - 1
            assert np.array_equal(rowval[a:b] - 1, oi) and np.array_equal(nzval[a:b], od)
        mask = L.unpack_bits(c.graph_edges_free(), nnz)
        es = rng.integers(0, nnz, size=200000)
        want = orc.unpack(orc.edges_free(w.X, rowval[es] - 1, cols[es] - 1, w.lohi, w.ss_lo, w.ss_hi), len(es))
        assert np.array_equal(mask[es], want)
        assert np.array_equal(c.points_free(), orc.points_free(w.X, w.lohi, w.ss_lo, w.ss_hi))
        # the STEP on the same world (what `bench.py --workload cfg3` times): half build with the edge tests fused into it (form 2: the
        # drain's broad phase in chains of six axes, flagged pairs through k_exact_pairs<12>, mask written by the ordering pass) -- the
        # same graph and the same mask, bit for bit, as the two-phase build and the whole sweep above
        whole_mask = c.graph_edges_free()
        c.set_option("rebuild_index", 1)
        for it in range(2):
            assert c.graph_step_device(w.r) == nnz
            assert c.stat("rdisc_half_used") == 1 and c.stat("sweep_form") == 2, (it, c.stat("sweep_form"))
            cp2, rv2, nz2, fr2 = _resident_graph(c, N)
            assert np.array_equal(cp2 + 1, colptr) and np.array_equal(rv2.astype(np.int64) + 1, rowval) and np.array_equal(nz2, nzval)
            assert np.array_equal(fr2.view(np.uint64), whole_mask)


def test_cfg4_double_integrator_full_size_properties(orc):
    """Kinodynamic FMT* graph, double integrator in R^4, N = 1e5 (BASELINE.json configs[3]): 1e10 pairs through the pilot-sized
    slot lists.  Column contract; the graph induced on a 2500-sample subset against the oracle's all-pairs steer of the
    subset (membership, cost, optimal time); sampled 5-waypoint edge bits against the oracle."""
    w = mp.workloads.cfg4()
    N = w.N
    m = w.X.shape[1] // 2
    with mp.Context(0) as c:
        c.upload_samples(w.X)
        c.upload_boxes(w.lohi, w.ss_lo, w.ss_hi)
        colptr, rowval, nzval, tval = c.di_graph(w.rho, w.r)
        nnz = len(rowval)
        deg = np.diff(colptr)
        assert colptr[0] == 1 and colptr[-1] == nnz + 1 and nnz > 5e7
        cols = np.repeat(np.arange(1, N + 1, dtype=np.int64), deg)
        assert not np.any(rowval == cols)
        dd = np.diff(rowval); same = cols[1:] == cols[:-1]
        assert np.all(dd[same] > 0)
        del dd, same
        assert nzval.max() <= w.r and nzval.min() > 0 and tval.min() > 0 and tval.max() <= w.r
        rng = np.random.default_rng(6)
        S = np.sort(rng.choice(N, size=2500, replace=False))
        oc, orow, oval, otv = orc.di_pairwise(w.X[S], w.rho, w.r)
        pos = np.full(N + 1, -1, dtype=np.int64); pos[S + 1] = np.arange(len(S))
        insub = (pos[rowval] >= 0) & (pos[cols] >= 0)
        gr, gc = pos[rowval[insub]], pos[cols[insub]]
        ocol = np.repeat(np.arange(len(S)), np.diff(oc))
        assert np.array_equal(gc, ocol) and np.array_equal(gr, orow)           # CSC order is preserved by the restriction
        assert np.array_equal(nzval[insub], oval) and np.array_equal(tval[insub], otv)
        mask, nseg = c.di_graph_edges_free()
        mask = L.unpack_bits(mask, nnz)
        es = rng.integers(0, nnz, size=3000)
        assert nseg.max() <= 4
        for e in es:
            fr = orc.di_is_free_motion(w.X[rowval[e] - 1], w.X[cols[e] - 1], w.rho, w.r, w.lohi, w.ss_lo, w.ss_hi)
            assert bool(mask[e]) == bool(fr), e
            assert (nseg[e] == 4) or not fr                                 # a free motion has passed all 4 segment tests


# ---- (e) the round-table sweep: packed rounds, pending lists, bounds shortcut ---------------------------------------------

@pytest.mark.parametrize("N,d,M,r,box_h,bounds", [
    (3000, 2, 25, 0.012, (0.02, 0.08), "inside"),     # degree ~1.3: a round holds four columns (or pieces of four)
    (3000, 2, 25, 0.05, (0.02, 0.08), "cut"),         # degree ~23: columns share quarters' rounds; some samples out of bounds
    (2500, 3, 256, 0.15, (0.03, 0.12), "inside"),     # 256 boxes = the kernel's limit (box ids fill a byte)
    (2500, 6, 0, 0.45, (0.1, 0.2), "cut"),            # no boxes: only the in_state_space bit
    (1500, 4, 120, 0.5, (0.15, 0.3), "inside"),       # degree ~250: columns over several rounds; dense boxes: lanes with > 4 pending
    (1500, 8, 60, 0.9, (0.2, 0.4), "none"),           # d = 8 (generic broad phase), no state-space bounds at all
])
def test_round_table_sweep_cases(orc, N, d, M, r, box_h, bounds):
    """k_graph_sweep_rt against the oracle's edge predicate over the corners of its design: rounds packed from several columns,
    columns over several rounds, the 256-box limit, more pending boxes per lane than its packed list holds, every sample in
    bounds (the per-row test is skipped), some out of bounds, no bounds; each with rows gathered from the cell-sorted copy and
    from the caller's array, and against the task-header kernel."""
    rng = np.random.default_rng(1000 + N + d + M)
    X, lohi = random_world(rng, N, d, M, *box_h)
    lo, hi = {"inside": (np.full(d, -0.5), np.full(d, 1.5)), "cut": (np.full(d, 0.03), np.full(d, 0.96)), "none": (None, None)}[bounds]
    oc, orow, oval = orc.rdisc_graph(X, r)
    want = orc.graph_edges_free(X, oc, orow, lohi, lo, hi)
    masks = []
    for rounds in (1, 0):
        with mp.Context(0) as c:
            c.set_option("sweep_rounds", rounds)
            c.upload_samples(X); c.upload_boxes(lohi, lo, hi)
            for _ in range(2):                                         # careful step, then the speculative one
                nnz = c.graph_step_device(r)
            colptr, rowval, nzval, free = _resident_graph(c, N)
            assert nnz == len(orow) and np.array_equal(colptr, oc) and np.array_equal(rowval, orow)
            masks.append(free.view(np.uint64).copy())
    for m in masks:
        assert np.array_equal(m, want)


def test_large_stress_of_the_timed_form_in_suite(orc):
    """tools/stress_large.py inside the suite: random worlds of 2e4 .. 2.5e5 samples in R^2 .. R^6 (clumps, density
    gradients, 0 .. 400 boxes) through the default step -- cold call, repeat, re-upload -- against scipy's kd-tree graph and the
    oracle's sweep of it, bit for bit; most of the steps must have run in the form bench.py times (half build, edge-test form 2)."""
    import os, sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import stress_large
    cases, edges, forms = stress_large.run(budget=100.0, seed=12, timed_form_only=True, max_pairs=4e7, sizes=(20000, 50000, 110000, 250000))
    steps = sum(forms.values())
    assert cases >= 3 and edges > 1e7, (cases, edges)
    assert forms.get((1, 2), 0) * 2 >= steps, forms


@pytest.mark.parametrize("d,N,deg", [(2, 30000, 4), (2, 20001, 2), (3, 40000, 0.15), (1, 5000, 4)])
def test_large_low_dimensional_world_takes_the_pipeline_with_the_exact_filter(orc, d, N, deg):
    """Worlds whose radius lies below the fp16 shell of globally normalised coordinates (the notebook's 2-D world scaled up,
    docs/MotionPlanning.ipynb cell 4): the step is still the single-pass pipeline -- half build, logs, edge tests fused (form 2) -- with
    the canonical fp64 test as the filter (k_rdisc_vf_w4).  Graph, costs and mask against the oracle over a cold step, a speculative
    repeat and new samples; duplicates and a ragged last tile included."""
    rng = np.random.default_rng(700 + d + N)
    r = float((deg / N) ** (1.0 / d) * 0.62) if d > 1 else 2e-5
    M = 40
    lo, hi = np.zeros(d), np.ones(d)
    lohi = mp.workloads.make_boxes(rng, M, d, 0.02, 0.12, [])
    X = rng.random((N, d)); X[77] = X[5]; X[N - 1] = X[N - 2]                  # exact duplicates: distance 0 is a neighbour
    X2 = rng.random((N, d))
    with mp.Context(0) as c:
        c.set_option("rebuild_index", 1)
        for Xi in (X, X, X2):
            c.upload_samples(Xi); c.upload_boxes(lohi, lo, hi)
            c.graph_step_device(r)
            assert c.stat("rdisc_path_used") == 2 and c.stat("filter_valu") == 1, (c.stat("rdisc_path_used"), c.stat("filter_valu"))
            assert c.stat("rdisc_half_used") == 1 and c.stat("sweep_form") == 2
            colptr, rowval, nzval, free = _resident_graph(c, N)
            oc, orow, oval = orc.rdisc_graph(Xi, r)
            assert np.array_equal(colptr, oc) and np.array_equal(rowval, orow) and np.array_equal(nzval, oval)
            assert np.array_equal(free.view(np.uint64), orc.graph_edges_free(Xi, oc, orow, lohi, lo, hi))
        assert c.stat("redo_count") == 0


def test_form_grid():
    """A cold ctx takes the timed form on its FIRST step, and keeps it: 2-, 3-, 4- and 6-dimensional worlds of 2e4 and
    1.1e5 uniform samples, 30 and 256 boxes, mean degree 6 and 60 -- cold step, repeat, new samples: EVERY world runs the pair-kernel
    pipeline (half build, edge-test form 2) and no build is redone because a capacity did not hold.  Where the radius lies below the
    fp16 shell of the matrix-core filter (2-D, N = 1.1e5 at degree 6: VERDICT r4 item 6) the filter is the exact fp64 one on the vector
    ALUs, the pipeline the same."""
    import os, sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import run_form_grid
    bad, valu = [], []
    for d, N, M, deg, nnz, out, flt, ms in run_form_grid.grid(sizes=(20000, 110000)):
        if flt: valu.append((d, N, deg))
        for it, (path, half, form, over, redone, why) in enumerate(out):
            if path != 2 or (half, form) != (1, 2) or over or redone:
                bad.append((d, N, M, deg, it, path, half, form, over, redone, why))
    assert not bad, bad
    assert (2, 110000, 6) in valu and all(d <= 3 for d, _, _ in valu), valu
