"""GPU parity tests (-m gpu) of the step bench.py TIMES, at the size it is timed at.

bench.py runs mpfmt_graph_step_device with rebuild_index = 1 on the north star (N = 1e6 in R^6, 200 boxes): half build of the r-disc
graph (every pair tested once, both columns' records written by the pair kernel) + the edge tests fused into it (broad phase in the pair
kernel's drain, flagged pairs through k_exact_pairs, mask written by the ordering pass).  At this size the oracle cannot build the
whole graph, so the resident CSC + mask are compared
  * with the ORACLE on sampled columns (KD-tree inball: rows and costs) and sampled edges (is_free_motion, parent first,
    src/planners/fmt.jl:75), and
  * entry for entry with the two-phase ABI forms (mpfmt_rdisc_count / _fill + mpfmt_graph_edges_free: the whole sweep) on a second
    ctx -- which tests/test_gpu_parity.py::test_north_star_full_size_properties compares with the oracle the same way,
on a cold first call, on repeats, and on NEW samples of the same (N, r) handed over from a device pointer
(mpfmt_upload_samples_device) -- the call sequence of the bench's timed loop.
"""
import numpy as np
import pytest

import motionplanning_jl_amd as mp
from test_gpu_parity import _resident_graph

pytestmark = pytest.mark.gpu
L = mp._lib


def _check_against_oracle(orc, X, r, lohi, ss_lo, ss_hi, colptr, rowval, nzval, free_words, rng, ncols=400, nedges=300000):
    N = len(X)
    nnz = int(colptr[-1])
    kd = orc.KDTree(X)
    for v in rng.integers(0, N, size=ncols):
        oi, od = kd.inball(int(v), r)
        a, b = int(colptr[v]), int(colptr[v + 1])
        assert np.array_equal(rowval[a:b], oi), "column %d: rows differ from the oracle" % v
        assert np.array_equal(nzval[a:b], od), "column %d: costs differ from the oracle" % v
    deg = np.diff(colptr)
    es = np.sort(rng.integers(0, nnz, size=nedges))
    cols = np.searchsorted(colptr, es, side="right") - 1
    assert np.all(colptr[cols] <= es) and np.all(es < colptr[cols + 1]) and deg.min() >= 0
    bits = L.unpack_bits(free_words.view(np.uint64), nnz)
    want = orc.unpack(orc.edges_free(X, rowval[es].astype(np.int64), cols.astype(np.int64), lohi, ss_lo, ss_hi), len(es))
    assert np.array_equal(bits[es], want), "sampled free bits differ from the oracle"


def test_north_star_step_path(orc):
    import torch
    w = mp.workloads.north_star()
    N = w.N
    rng = np.random.default_rng(41)
    # reference of the whole arrays: the two-phase forms on a second ctx (the whole sweep kernel)
    with mp.Context(0) as b:
        b.upload_samples(w.X); b.upload_boxes(w.lohi, w.ss_lo, w.ss_hi)
        colptr_b, rowval_b, nzval_b = b.rdisc_graph(w.r)
        mask_b = b.graph_edges_free()
    nnz_ref = len(rowval_b)
    assert nnz_ref == 107492200                               # (the workload stream is pinned: tests/golden/stream_heads.json)
    colptr_b = colptr_b - 1
    rowval_b = (rowval_b - 1).astype(np.int32)
    with mp.Context(0) as c:
        c.set_option("rebuild_index", 1)
        c.upload_samples(w.X); c.upload_boxes(w.lohi, w.ss_lo, w.ss_hi)
        forms = []
        for it in range(3):                                   # cold, then two calls that take the previous sizes on trust
            nnz = c.graph_step_device(w.r)
            forms.append((c.stat("rdisc_path_used"), c.stat("rdisc_half_used"), c.stat("sweep_form")))
            assert nnz == nnz_ref, it
            colptr, rowval, nzval, free = _resident_graph(c, N)
            assert np.array_equal(colptr, colptr_b), it
            assert np.array_equal(rowval, rowval_b), it
            assert np.array_equal(nzval, nzval_b), it
            assert np.array_equal(free.view(np.uint64), mask_b), it
            _check_against_oracle(orc, w.X, w.r, w.lohi, w.ss_lo, w.ss_hi, colptr, rowval, nzval, free, rng)
            if it == 2:
                # the drop-in precompute! of julia/MPFmtHIP.jl: the same arrays through mpfmt_graph_export (1-based Int64, page-locked destinations)
                ec, er, ev, em, rate = c.graph_export(pinned=True)
                assert np.array_equal(ec - 1, colptr) and np.array_equal(er - 1, rowval) and np.array_equal(ev, nzval)
                assert np.array_equal(em, free.view(np.uint64))
                assert rate > 3.0, rate                       # GB/s (PCIe Gen4 x16 would give ~25, Gen5 ~50; pageable copies ~3)
                print("mpfmt_graph_export: %.1f GB/s" % rate)
                del ec, er, ev, em
            del colptr, rowval, nzval, free
        assert forms[1] == (2, 1, 2) and forms[2] == (2, 1, 2), forms      # the timed form: MFMA pair kernel, half build, fused edge tests
        assert forms[0] == (2, 1, 2), forms                                # ... and the cold call takes it too
        del rowval_b, nzval_b, mask_b, colptr_b
        # new samples of the same (N, r), resident in HBM: what a planner's next call looks like (and bench.py's timed loop)
        for seed in (101, 102):
            X2 = np.random.default_rng(seed).random((N, w.d))
            X2[0] = w.X[0]; X2[-1] = w.X[-1]
            t = torch.from_numpy(X2).to("cuda:0")
            torch.cuda.synchronize()
            c.upload_samples_device(t.data_ptr(), N, w.d)
            nnz = c.graph_step_device(w.r)
            assert (c.stat("rdisc_path_used"), c.stat("rdisc_half_used"), c.stat("sweep_form")) == (2, 1, 2), seed
            colptr, rowval, nzval, free = _resident_graph(c, N)
            assert nnz == colptr[-1]
            _check_against_oracle(orc, X2, w.r, w.lohi, w.ss_lo, w.ss_hi, colptr, rowval, nzval, free, rng, ncols=300, nedges=200000)
            # symmetry of the whole graph: sum and xor checksums of the directed keys and of their transposes
            cols = np.repeat(np.arange(N, dtype=np.int64), np.diff(colptr))
            k1 = cols * N + rowval
            k2 = rowval.astype(np.int64) * N + cols
            assert int(k1.sum()) == int(k2.sum()) and int(np.bitwise_xor.reduce(k1)) == int(np.bitwise_xor.reduce(k2))
            d = np.diff(rowval.astype(np.int64))
            assert np.all(d[cols[1:] == cols[:-1]] > 0)                          # ascending inside every column
            del cols, k1, k2, d, colptr, rowval, nzval, free, t


def test_clustered_samples_at_the_north_star_size(orc):
    """Throughput under NON-UNIFORM samples is only worth reporting if the timed form survives them: the north star's
    world with 30 % of the samples in a Gaussian cluster (workloads.north_star_clustered: density ~5.7 x uniform at the centre, columns several
    times the mean).  The cold call may have to redo its build (the log capacity of a cold ctx assumes a uniform density); from the SECOND
    step on -- repeats and NEW clustered sample sets alike -- the step must run as (matrix-core pair kernel, half build, fused edge tests)
    with no build redone.  Graph and mask against the oracle on sampled columns / edges, the densest columns included."""
    import torch
    w = mp.workloads.north_star_clustered()
    N = w.N
    rng = np.random.default_rng(43)
    with mp.Context(0) as c:
        c.set_option("rebuild_index", 1)
        c.upload_samples(w.X); c.upload_boxes(w.lohi, w.ss_lo, w.ss_hi)
        redo = []
        for it in range(3):
            nnz = c.graph_step_device(w.r)
            redo.append(c.stat("redo_count"))
            form = (c.stat("rdisc_path_used"), c.stat("rdisc_half_used"), c.stat("sweep_form"))
            if it >= 1:
                assert form == (2, 1, 2), (it, form)
                assert redo[it] == redo[it - 1], (it, redo, c.stat("redo_reason"))
            if it != 1:
                colptr, rowval, nzval, free = _resident_graph(c, N)
                assert nnz == colptr[-1] and nnz > 1.2 * 107492200        # (the cluster adds edges: the density enters squared)
                deg = np.diff(colptr)
                assert deg.max() > 4 * deg.mean()                          # ... and long columns
                _check_against_oracle(orc, w.X, w.r, w.lohi, w.ss_lo, w.ss_hi, colptr, rowval, nzval, free, rng, ncols=200, nedges=200000)
                kd = orc.KDTree(w.X)
                for v in np.argsort(deg)[-20:]:                            # the longest columns
                    oi, od = kd.inball(int(v), w.r)
                    a, b = int(colptr[v]), int(colptr[v + 1])
                    assert np.array_equal(rowval[a:b], oi) and np.array_equal(nzval[a:b], od), v
                del colptr, rowval, nzval, free, kd
        for k in (1, 2):                                                   # new clustered sample sets of the same (N, r): bench.py's timed loop
            X2 = mp.workloads.resample(w, k)
            t = torch.from_numpy(X2).to("cuda:0")
            torch.cuda.synchronize()
            c.upload_samples_device(t.data_ptr(), N, w.d)
            nnz = c.graph_step_device(w.r)
            assert (c.stat("rdisc_path_used"), c.stat("rdisc_half_used"), c.stat("sweep_form")) == (2, 1, 2), k
            assert c.stat("redo_count") == redo[-1], (k, c.stat("redo_count"), redo, c.stat("redo_reason"))
            if k == 2:
                colptr, rowval, nzval, free = _resident_graph(c, N)
                assert nnz == colptr[-1]
                _check_against_oracle(orc, X2, w.r, w.lohi, w.ss_lo, w.ss_hi, colptr, rowval, nzval, free, rng, ncols=200, nedges=200000)
                del colptr, rowval, nzval, free
            del t


def test_goal_biased_free_samples_at_the_north_star_size(orc):
    """The sample sets the library's own sampler emits (mpfmt_sample_free_biased, src/sampling.jl:11-45: free space only, goal bias) through the timed
    step at the north star's N and r: ~500 samples inside the goal ball (columns of ~600 entries against a mean of ~120), none inside an
    obstacle.  From the second step on -- repeats and NEW biased sets -- (pair kernel, half build, fused edge tests) and no build redone;
    graph and mask against the oracle on sampled columns, the goal ball's columns included."""
    w = mp.workloads.north_star_biased()
    N = w.N
    rng = np.random.default_rng(44)
    with mp.Context(0) as c:
        c.set_option("rebuild_index", 1)
        c.upload_samples(w.X); c.upload_boxes(w.lohi, w.ss_lo, w.ss_hi)
        X = w.device_set(c, 0)
        assert orc.unpack(orc.points_free(X[::97], w.lohi, w.ss_lo, w.ss_hi), len(X[::97])).all()        # free space only
        ingoal = np.flatnonzero(np.linalg.norm(X - w.goal_center, axis=1) <= w.goal_radius)
        assert 300 < len(ingoal) < 900, len(ingoal)
        redo = []
        for it in range(3):
            nnz = c.graph_step_device(w.r)
            redo.append(c.stat("redo_count"))
            if it >= 1:
                assert (c.stat("rdisc_path_used"), c.stat("rdisc_half_used"), c.stat("sweep_form")) == (2, 1, 2), it
                assert redo[it] == redo[it - 1], (it, redo, c.stat("redo_reason"))
        colptr, rowval, nzval, free = _resident_graph(c, N)
        assert nnz == colptr[-1]
        deg = np.diff(colptr)
        assert deg[ingoal].mean() > 2 * deg.mean()
        _check_against_oracle(orc, X, w.r, w.lohi, w.ss_lo, w.ss_hi, colptr, rowval, nzval, free, rng, ncols=200, nedges=200000)
        kd = orc.KDTree(X)
        for v in ingoal[:25]:
            oi, od = kd.inball(int(v), w.r)
            a, b = int(colptr[v]), int(colptr[v + 1])
            assert np.array_equal(rowval[a:b], oi) and np.array_equal(nzval[a:b], od), v
        del colptr, rowval, nzval, free, kd
        for k in (1, 2):                                                   # new biased sets (the sampler leaves each uploaded)
            X2 = w.device_set(c, k)
            nnz = c.graph_step_device(w.r)
            assert (c.stat("rdisc_path_used"), c.stat("rdisc_half_used"), c.stat("sweep_form")) == (2, 1, 2), k
            assert c.stat("redo_count") == redo[-1], (k, c.stat("redo_count"), redo, c.stat("redo_reason"))
        colptr, rowval, nzval, free = _resident_graph(c, N)
        _check_against_oracle(orc, X2, w.r, w.lohi, w.ss_lo, w.ss_hi, colptr, rowval, nzval, free, rng, ncols=150, nedges=150000)


def test_upload_samples_device_equals_host_upload(orc):
    """mpfmt_upload_samples_device against mpfmt_upload_samples: same bounding box (hence the same grid), same graph and mask;
    a non-finite coordinate is refused."""
    import torch
    rng = np.random.default_rng(5)
    N, d, r = 30011, 3, 0.06
    X = rng.random((N, d)) * np.array([1.0, 0.5, 2.0]) - 0.25
    lohi = mp.workloads.make_boxes(rng, 25, d, 0.05, 0.2, [])
    lo, hi = np.full(d, -0.3), np.full(d, 2.0)
    got = []
    for dev in (False, True):
        with mp.Context(0) as c:
            c.upload_boxes(lohi, lo, hi)
            if dev:
                t = torch.from_numpy(X).to("cuda:0"); torch.cuda.synchronize()
                c.upload_samples_device(t.data_ptr(), N, d)
            else:
                c.upload_samples(X)
            c.graph_step_device(r)
            got.append(_resident_graph(c, N) + (c.graph_stats()["cells"],))
    for u, v in zip(*got):
        assert np.array_equal(u, v)
    oc, orow, oval = orc.rdisc_graph(X, r)
    assert np.array_equal(got[1][0], oc) and np.array_equal(got[1][1], orow) and np.array_equal(got[1][2], oval)
    assert np.array_equal(got[1][3].view(np.uint64), orc.graph_edges_free(X, oc, orow, lohi, lo, hi))
    X[123, 1] = np.nan
    with mp.Context(0) as c:
        t = torch.from_numpy(X).to("cuda:0"); torch.cuda.synchronize()
        with pytest.raises(mp.MPFMTError):
            c.upload_samples_device(t.data_ptr(), N, d)
        # the host upload checks on the device too: the error names the sample (1-based), the ctx keeps what it had (here: nothing), and a
        # good set uploaded afterwards is served as if nothing had happened
        c.upload_boxes(lohi, lo, hi)
        with pytest.raises(mp.MPFMTError, match="sample 124 "):
            c.upload_samples(X)
        with pytest.raises(mp.MPFMTError, match="no samples"):                   # (the ctx never accepted a set)
            c.graph_step_device(r)
        X[123, 1] = 0.5; X[7, 2] = np.inf
        with pytest.raises(mp.MPFMTError, match="sample 8 "):
            c.upload_samples(X)
        X[7, 2] = 0.25
        c.upload_samples(X); c.graph_step_device(r)
        oc, orow, oval = orc.rdisc_graph(X, r)
        colptr, rowval, nzval, free = _resident_graph(c, N)
        assert np.array_equal(colptr, oc) and np.array_equal(rowval, orow) and np.array_equal(nzval, oval)
        # a refused set leaves the ctx as it was: the resident graph stays readable, and the next step is the step of the OLD samples
        Xbad = X.copy(); Xbad[N - 1, 0] = -np.inf
        with pytest.raises(mp.MPFMTError, match="sample %d " % N):
            c.upload_samples(Xbad)
        tb = torch.from_numpy(Xbad).to("cuda:0"); torch.cuda.synchronize()
        with pytest.raises(mp.MPFMTError):
            c.upload_samples_device(tb.data_ptr(), N, d)
        c2, r2, n2, f2 = _resident_graph(c, N)
        assert np.array_equal(c2, colptr) and np.array_equal(r2, rowval) and np.array_equal(n2, nzval) and np.array_equal(f2, free)
        c.set_option("rebuild_index", 1)
        assert c.graph_step_device(r) == len(orow)
        c2, r2, n2, f2 = _resident_graph(c, N)
        assert np.array_equal(c2, oc) and np.array_equal(r2, orow) and np.array_equal(n2, oval) and np.array_equal(f2, free)


def test_four_million_samples_step(orc):
    """Four times the north star (N = 4e6 in R^6, r from the fmt.jl:39 rule, 5.0e8 directed edges, 6 GB of CSC): the timed step's form on a
    cold ctx, sampled columns of the resident graph against the oracle's KD-tree, sampled free bits against the oracle's edge predicate
    (only the sampled slices cross PCIe)."""
    import torch
    from motionplanning_jl_amd.distributed import DevArray
    w = mp.workloads.north_star(4_000_000)
    N = w.N
    rng = np.random.default_rng(77)
    with mp.Context(0) as c:
        c.set_option("rebuild_index", 1)
        c.upload_samples(w.X); c.upload_boxes(w.lohi, w.ss_lo, w.ss_hi)
        nnz = c.graph_step_device(w.r)
        assert (c.stat("rdisc_path_used"), c.stat("rdisc_half_used"), c.stat("sweep_form")) == (2, 1, 2)
        assert c.stat("redo_count") == 0
        assert 4.5e8 < nnz < 5.5e8 and nnz % 2 == 0
        cp, rv, nz, fr = c.graph_device_ptrs()
        torch.cuda.synchronize()
        colptr = torch.as_tensor(DevArray(cp, N + 1, "<i8"), device="cuda:0").cpu().numpy()
        assert colptr[0] == 0 and colptr[-1] == nnz and np.all(np.diff(colptr) >= 0)
        rowval_d = torch.as_tensor(DevArray(rv, nnz, "<i4"), device="cuda:0")
        nzval_d = torch.as_tensor(DevArray(nz, nnz, "<f8"), device="cuda:0")
        free_d = torch.as_tensor(DevArray(fr, (nnz + 63) // 64, "<i8"), device="cuda:0")
        kd = orc.KDTree(w.X)
        for v in rng.integers(0, N, size=250):
            oi, od = kd.inball(int(v), w.r)
            a, b = int(colptr[v]), int(colptr[v + 1])
            rows = rowval_d[a:b].cpu().numpy()
            assert np.array_equal(rows, oi), v
            assert np.array_equal(nzval_d[a:b].cpu().numpy(), od), v
            want = orc.unpack(orc.edges_free(w.X, oi, np.full(len(oi), v), w.lohi, w.ss_lo, w.ss_hi), len(oi))
            words = free_d[a // 64:(b + 63) // 64 + 1].cpu().numpy().view(np.uint64)
            bits = L.unpack_bits(words, len(words) * 64)[a - (a // 64) * 64:][:b - a]
            assert np.array_equal(bits, want), v
        # a second step on the first one's sizes (speculative): same graph size, same form, nothing redone
        assert c.graph_step_device(w.r) == nnz and c.stat("redo_count") == 0 and c.stat("sweep_form") == 2
