"""GPU tests (-m gpu) of the device-resident wavefront FMT* driver (mpfmt_fmtstar_wavefront, mpfmt_wf_*) and of the RCCL
exchange behind the C ABI (mpfmt_comm_*, mpfmt_allgather_free_mask).

Parity bar:
  * single-node batches: tree, costs, path, collision_checks and the final node equal the sequential recursion
    (oracle orc_fmtstar = src/planners/fmt.jl:3-119) exactly;
  * cost-band batches: every step equals the batch form of the loop body (oracle orc_expand) on the same (W, H, C), and
    the whole solve equals the oracle's batched loop (orc_fmt_wavefront_graph) exactly; the cost is reported against the
    sequential one;
  * sharded: G shards on one GPU with the exchange made by hand (wf_triples / wf_commit) give the unsharded result; the
    RCCL path runs with a 1-rank communicator.
"""
import numpy as np
import pytest

import motionplanning_jl_amd as mp

pytestmark = pytest.mark.gpu
L = mp._lib


@pytest.fixture(scope="module")
def ctx():
    c = mp.Context(0)
    yield c
    c.close()


def world(N, d, M, seed, goal_radius=0.1, h=(0.05, 0.12)):
    return mp.workloads.make("t", N, d, M, h[0], h[1], seed=seed, goal_radius=goal_radius)


def upload(ctx, w):
    ctx.upload_samples(w.X)
    ctx.upload_boxes(w.lohi, w.ss_lo, w.ss_hi)


def same_solution(a, b, tree=True):
    assert a["status"] == b["status"] and a["z"] == b["z"]
    assert a["collision_checks"] == b["collision_checks"]
    assert a["cost"] == b["cost"]
    assert np.array_equal(a["path"], b["path"])
    if tree:
        assert np.array_equal(a["A"], b["A"]) and np.array_equal(a["C"], b["C"])


@pytest.mark.parametrize("N,d,M,seed", [(1000, 2, 20, 1), (3000, 2, 25, 11), (5000, 3, 40, 2), (8000, 6, 100, 3)])
def test_single_node_batches_equal_the_sequential_loop(ctx, orc, N, d, M, seed):
    w = world(N, d, M, seed)
    upload(ctx, w)
    seq = ctx.fmtstar(w.r, L.GOAL_BALL, w.goal_params())
    ref = orc.fmtstar(w.X, w.r, orc.GOAL_BALL, w.goal_params(), w.lohi, w.ss_lo, w.ss_hi, nn_mode=1)
    for eager in (False, True):
        got = ctx.fmtstar_wavefront(w.r, L.GOAL_BALL, w.goal_params(), single=True, eager=eager)
        same_solution(got, seq)
        assert got["status"] == ref["status"] and got["collision_checks"] == ref["collision_checks"]
        assert np.array_equal(got["A"] - 1, ref["A"]) and np.array_equal(got["path"] - 1, ref["path"])
        assert np.array_equal(got["C"], ref["C"])
        assert got["z"] - 1 == ref["z"]


def test_single_node_batches_unreachable_goal(ctx, orc):
    """The open set runs dry (fmt.jl:85-89 break): status failed, z = the last dequeued node."""
    rng = np.random.default_rng(5)
    X = rng.random((1500, 2)) * 0.45          # nothing near the goal
    X[0] = 0.1
    lohi = np.array([[[0.2, 0.2], [0.25, 0.3]]])
    ctx.upload_samples(X)
    ctx.upload_boxes(lohi, np.zeros(2), np.ones(2))
    g = [0.9, 0.9, 0.05]
    ref = orc.fmtstar(X, 0.05, orc.GOAL_BALL, g, lohi, np.zeros(2), np.ones(2), nn_mode=1)
    got = ctx.fmtstar_wavefront(0.05, L.GOAL_BALL, g, single=True)
    assert got["status"] == 0 == ref["status"]
    assert got["z"] - 1 == ref["z"] and got["cost"] == ref["cost"] and got["collision_checks"] == ref["collision_checks"]
    assert np.array_equal(got["A"] - 1, ref["A"]) and np.array_equal(got["path"] - 1, ref["path"])
    # banded: ends the same way (the set of connected samples may differ: a blocked best parent is retried at other times)
    gb = ctx.fmtstar_wavefront(0.05, L.GOAL_BALL, g, band=0.02)
    assert gb["status"] == 0 and (gb["A"] > 0).sum() > 100


def test_infeasible_init_and_errors(ctx):
    X = np.array([[0.1, 0.1], [0.5, 0.5], [0.9, 0.9]])
    ctx.upload_samples(X)
    ctx.upload_boxes(np.array([[[0.0, 0.0], [0.2, 0.2]]]), np.zeros(2), np.ones(2))
    with pytest.raises(mp.MPFMTError) as e:
        ctx.fmtstar_wavefront(0.7, L.GOAL_BALL, [0.9, 0.9, 0.05])
    assert e.value.code == L.ERR_INFEASIBLE
    with pytest.raises(mp.MPFMTError) as e:
        ctx.wf_step()
    assert e.value.code == L.ERR_STATE
    ctx.upload_boxes(np.zeros((0, 2, 2)), np.zeros(2), np.ones(2))
    with pytest.raises(mp.MPFMTError) as e:
        ctx.fmtstar_wavefront(0.7, L.GOAL_BALL, [0.9, 0.9, 0.05], band=-1.0)
    assert e.value.code == L.ERR_ARG


@pytest.mark.parametrize("N,d,M,seed,bandf", [(900, 2, 20, 1, 0.5), (1200, 3, 30, 2, 1.0), (1500, 6, 60, 3, 0.25)])
def test_band_steps_equal_oracle_expand(ctx, orc, N, d, M, seed, bandf):
    """Every step against the oracle's batch form of the loop body on the same sets (brute force over all samples)."""
    w = world(N, d, M, seed)
    upload(ctx, w)
    F = ctx.points_free()
    ctx.wf_begin(w.r, L.GOAL_BALL, w.goal_params(), band=bandf * w.r)
    steps = 0
    while True:
        W0, H0, C0, A0 = ctx.wf_state()
        info = ctx.wf_step()
        zs = ctx.wf_batch()
        W1, H1, C1, A1 = ctx.wf_state()
        if info["done"] == 2:
            assert zs.size == 0 and not L.unpack_bits(H0, N).any()
            break
        Hb = L.unpack_bits(H0, N)
        cmin = C0[Hb].min()
        assert info["cmin"] == cmin
        assert np.array_equal(np.sort(zs) - 1, np.flatnonzero(Hb & (C0 <= cmin + bandf * w.r)))
        if info["done"] == 1:
            assert np.array_equal(W1, W0) and np.array_equal(C1, C0)        # the goal batch is not expanded
            break
        oxs, oym, ocm, ofr = orc.expand(w.X, w.r, W0, H0, F, C0, zs - 1, w.lohi, w.ss_lo, w.ss_hi)
        assert info["nx"] == len(oxs)
        conn = oxs[ofr]
        Wb0, Wb1 = L.unpack_bits(W0, N), L.unpack_bits(W1, N)
        assert np.array_equal(np.flatnonzero(Wb0 & ~Wb1), conn)             # exactly the free ones left W
        assert np.array_equal(A1[conn] - 1, oym[ofr]) and np.array_equal(C1[conn], ocm[ofr])
        rest = np.ones(N, bool); rest[conn] = False
        assert np.array_equal(A1[rest], A0[rest]) and np.array_equal(C1[rest], C0[rest])
        Hexp = Hb.copy(); Hexp[zs - 1] = False; Hexp[conn] = True            # fmt.jl:83-84 for the batch
        assert np.array_equal(L.unpack_bits(H1, N), Hexp)
        assert info["nconn"] == len(conn)
        steps += 1
    assert steps > 2
    res = ctx.wf_finish()
    assert res["status"] == (1 if info["done"] == 1 else 0)


@pytest.mark.parametrize("N,d,M,seed", [(4000, 2, 25, 7), (20000, 3, 60, 8), (30000, 6, 200, 9)])
def test_band_solve_equals_oracle_wavefront(ctx, orc, N, d, M, seed):
    w = world(N, d, M, seed)
    upload(ctx, w)
    colptr, rowval, nzval = ctx.rdisc_graph(w.r)
    F = ctx.points_free()
    seq = orc.fmtstar_graph(w.X, colptr - 1, rowval - 1, nzval, None, F, orc.GOAL_BALL, w.goal_params(), w.lohi, w.ss_lo, w.ss_hi)
    for bandf in (0.0, 0.1, 0.5, 2.0):
        ref = orc.fmt_wavefront_graph(w.X, colptr - 1, rowval - 1, nzval, None, F, orc.GOAL_BALL, w.goal_params(), w.lohi,
                                      w.ss_lo, w.ss_hi, band=bandf * w.r)
        for eager in (False, True):
            got = ctx.fmtstar_wavefront(w.r, L.GOAL_BALL, w.goal_params(), band=bandf * w.r, eager=eager)
            assert got["status"] == ref["status"] and got["z"] - 1 == ref["z"]
            assert got["collision_checks"] == ref["collision_checks"]
            assert np.array_equal(got["A"] - 1, ref["A"]) and np.array_equal(got["C"], ref["C"])
            assert np.array_equal(got["path"] - 1, ref["path"])
            assert got["info"]["iters"] == ref["iters"]
        if seq["status"] == 1:
            assert ref["status"] == 1
            assert ref["cost"] >= seq["cost"] * (1 - 1e-12)      # the sequential order is the best FMT* order
            assert ref["cost"] <= seq["cost"] * 1.25
    # no checkpts (fmt.jl:8 checkpts = false)
    ref = orc.fmt_wavefront_graph(w.X, colptr - 1, rowval - 1, nzval, None, None, orc.GOAL_BALL, w.goal_params(), w.lohi, w.ss_lo,
                                  w.ss_hi, band=0.3 * w.r)
    got = ctx.fmtstar_wavefront(w.r, L.GOAL_BALL, w.goal_params(), band=0.3 * w.r, checkpts=False)
    assert np.array_equal(got["A"] - 1, ref["A"]) and got["collision_checks"] == ref["collision_checks"]


def test_position_space_from_the_second_solve_on_a_graph(orc):
    """The device solve keeps its sets a second time by cell-sorted position once a graph is solved on AGAIN (the entries' positions cost a
    pass over the whole graph, which a first -- cold -- solve does not pay): first solve by caller index, later ones by position, a new
    sample set starts over; options 0 / 2 pin either form.  Every solve equals the oracle's batched loop exactly."""
    w = world(30000, 6, 150, 21)
    w2 = world(30000, 6, 150, 22)
    with mp.Context(0) as c:
        for opt, want in ((1, (0, 1, 1)), (2, (1, 1, 1)), (0, (0, 0, 0))):
            c.set_option("wf_pos_space", opt)
            for wi in (w, w2):
                upload(c, wi)
                ref = None
                for k in range(3):
                    got = c.fmtstar_wavefront(wi.r, L.GOAL_BALL, wi.goal_params(), band=0.25 * wi.r)
                    assert c.stat("wf_pos_space_used") == want[k], (opt, k)
                    if ref is None:
                        colptr, rowval, nzval = c.graph_export()[:3]
                        ref = orc.fmt_wavefront_graph(wi.X, colptr - 1, rowval - 1, nzval, None, c.points_free(), orc.GOAL_BALL, wi.goal_params(),
                                                      wi.lohi, wi.ss_lo, wi.ss_hi, band=0.25 * wi.r)
                    assert got["status"] == ref["status"] and got["z"] - 1 == ref["z"] and got["collision_checks"] == ref["collision_checks"]
                    assert np.array_equal(got["A"] - 1, ref["A"]) and np.array_equal(got["C"], ref["C"]) and np.array_equal(got["path"] - 1, ref["path"])


def test_wavefront_in_the_sat2d_world(ctx, orc):
    """Checkers without a lane-per-obstacle form answer from the swept mask: same result as the host recursion."""
    rng = np.random.default_rng(3)
    N = 3000
    X = rng.random((N, 2)); X[0] = 0.05; X[-1] = 0.95
    shapes = [("circle", (0.5, 0.5), 0.15), ("polygon", [(0.2, 0.6), (0.35, 0.6), (0.35, 0.9), (0.2, 0.9)]),
              ("polygon", [(0.6, 0.1), (0.8, 0.15), (0.7, 0.35)])]
    ctx.upload_samples(X)
    ctx.upload_shapes2d(shapes, np.zeros(2), np.ones(2))
    g = [0.95, 0.95, 0.05]
    seq = ctx.fmtstar(0.06, L.GOAL_BALL, g)
    got = ctx.fmtstar_wavefront(0.06, L.GOAL_BALL, g, single=True)
    same_solution(got, seq)
    ctx.upload_boxes(np.zeros((0, 2, 2)), np.zeros(2), np.ones(2))


@pytest.mark.parametrize("G", [2, 3])
def test_sharded_wavefront_manual_exchange(orc, G):
    """G shards on one GPU, the per-wavefront exchange made by the caller: same tree as the unsharded solve."""
    w = world(6000, 3, 40, 21)
    band = 0.4 * w.r
    with mp.Context(0) as c0:
        upload(c0, w)
        want = c0.fmtstar_wavefront(w.r, L.GOAL_BALL, w.goal_params(), band=band)
    cs = [mp.Context(0) for _ in range(G)]
    try:
        for g, c in enumerate(cs):
            c.set_shard(g, G)
            upload(c, w)
            c.wf_begin(w.r, L.GOAL_BALL, w.goal_params(), band=band)
        steps = 0
        while True:
            infos = [c.wf_step() for c in cs]
            assert len({i["done"] for i in infos}) == 1 and len({i["nz"] for i in infos}) == 1
            if infos[0]["done"]:
                break
            trips = [c.wf_triples() for c in cs]
            x = np.concatenate([t[0] for t in trips]); y = np.concatenate([t[1] for t in trips]); cc = np.concatenate([t[2] for t in trips])
            assert len(np.unique(x)) == len(x)                         # every sample has one owner
            for c in cs:
                c.wf_commit(x, y, cc)
            steps += 1
        res = [c.wf_finish() for c in cs]
        checks = sum(r["collision_checks"] for r in res)
        for r in res:
            assert r["status"] == want["status"] and r["z"] == want["z"] and r["cost"] == want["cost"]
            assert np.array_equal(r["A"], want["A"]) and np.array_equal(r["C"], want["C"]) and np.array_equal(r["path"], want["path"])
        assert checks == want["collision_checks"]
        assert steps + 1 == want["info"]["iters"]
    finally:
        for c in cs:
            c.close()


def test_rccl_one_rank_communicator(orc):
    """The RCCL calls behind the ABI with a 1-rank communicator: mask gather (steady state and growth) and the wavefront
    exchange (forced through the sharded code path)."""
    import torch
    w = world(20000, 3, 40, 31)
    with mp.Context(0) as c:
        c.comm_create(0, 1, mp._lib.comm_unique_id())
        upload(c, w)
        for rr in (w.r, w.r, 0.7 * w.r, 1.6 * w.r, w.r):              # same size, shrink (< half), grow beyond the capacity
            nnz = c.graph_step_device(rr)
            _, _, _, fptr = c.graph_device_ptrs()
            ptr, stride, words, nnzs = c.allgather_free_mask(1)
            assert words[0] == (nnz + 63) // 64 and nnzs[0] == nnz and stride >= words[0] + 2
            got = torch.as_tensor(mp.distributed.DevArray(ptr, stride), device="cuda:0").cpu().numpy()
            want = torch.as_tensor(mp.distributed.DevArray(fptr, words[0]), device="cuda:0").cpu().numpy()
            assert got[0] == words[0] and got[1] == nnz
            assert np.array_equal(got[2:2 + words[0]], want)
            assert not got[2 + words[0]:].any()                         # zero padded
        # split form: the gather overlaps the next build
        c.graph_step_device(w.r)
        c.allgather_free_mask_launch()
        with pytest.raises(mp.MPFMTError):
            c.allgather_free_mask_launch()
        c.allgather_free_mask_finish(1)
        want = c.fmtstar_wavefront(w.r, L.GOAL_BALL, w.goal_params(), band=0.3 * w.r)
        c.set_option("wf_force_sharded", 1)
        got = c.fmtstar_wavefront(w.r, L.GOAL_BALL, w.goal_params(), band=0.3 * w.r)
        c.set_option("wf_force_sharded", 0)
        same_solution(got, want)
        c.comm_destroy()
        c.comm_destroy()                                                 # idempotent


def test_north_star_wavefront_solve(orc):
    """N = 1e6, R^6, 200 boxes: the whole solve on the device.  Checks the tree invariants the loop guarantees, sampled
    edges against the oracle, and the cost against the sequential recursion on the same graph."""
    w = mp.workloads.north_star()
    N = w.N
    with mp.Context(0) as c:
        upload(c, w)
        seq = c.fmtstar(w.r, L.GOAL_BALL, w.goal_params())
        for bandf in (0.1, 0.5):
            got = c.fmtstar_wavefront(w.r, L.GOAL_BALL, w.goal_params(), band=bandf * w.r)
            assert got["status"] == 1 == seq["status"]
            A, Cc = got["A"], got["C"]
            kids = np.flatnonzero(A > 0)
            par = A[kids] - 1
            dist = np.sqrt(((w.X[kids] - w.X[par]) ** 2).sum(1))
            assert dist.max() <= w.r * (1 + 1e-12)
            assert np.allclose(Cc[kids], Cc[par] + dist, rtol=1e-12, atol=0)
            rng = np.random.default_rng(0)
            es = rng.choice(len(kids), size=200000, replace=False)
            fr = orc.unpack(orc.edges_free(w.X, par[es], kids[es], w.lohi, w.ss_lo, w.ss_hi), len(es))
            assert fr.all()                                             # every tree edge is collision free
            assert got["path"][0] == 1 and got["path"][-1] == got["z"]
            assert orc.is_goal_pt(w.X[got["z"] - 1], orc.GOAL_BALL, w.goal_params())
            assert got["cost"] >= seq["cost"] * (1 - 1e-12) and got["cost"] <= seq["cost"] * 1.10
            print("north star wavefront band %.2f r: cost %.6f (sequential %.6f), %d steps, %d checks (sequential %d), loop %.1f ms"
                  % (bandf, got["cost"], seq["cost"], got["info"]["iters"], got["collision_checks"], seq["collision_checks"], got["ms_host_loop"]))


@pytest.mark.parametrize("N,d,M", [(1, 2, 3), (2, 2, 0), (63, 1, 2), (64, 3, 0), (65, 6, 300), (130, 12, 70), (200, 16, 5), (700, 2, 64)])
def test_tiny_and_ragged_worlds(ctx, orc, N, d, M):
    """Ragged sizes (N around the 64-bit mask words, one sample, no obstacles, more obstacles than one lane round, d up to 16):
    single-node and banded solves against the oracle's loops; the init may already be a goal node."""
    rng = np.random.default_rng(1000 + N + d)
    X = rng.random((N, d))
    X[0] = 0.1
    if N > 1:
        X[-1] = 0.9
    c = rng.random((M, d)); h = 0.02 + 0.05 * rng.random((M, d))
    lohi = np.stack([c - h, c + h], axis=1) if M else np.zeros((0, 2, d))
    keep = np.array([not (np.all((lo <= X[0]) & (X[0] <= hi))) for lo, hi in lohi], dtype=bool) if M else np.zeros(0, bool)
    lohi = lohi[keep] if M else lohi
    lo, hi = np.zeros(d), np.ones(d)
    r = 2.5 * (1.0 / max(N, 2)) ** (1.0 / d) if d <= 6 else 1.2
    for goal in (np.concatenate([np.full(d, 0.9), [0.12]]), np.concatenate([np.full(d, 0.1), [0.05]])):       # far goal / the init itself
        ctx.upload_samples(X); ctx.upload_boxes(lohi, lo, hi)
        colptr, rowval, nzval = ctx.rdisc_graph(r)
        F = ctx.points_free()
        for single, bandf in ((True, 0.0), (False, 0.0), (False, 0.7)):
            ref = orc.fmt_wavefront_graph(X, colptr - 1, rowval - 1, nzval, None, F, orc.GOAL_BALL, goal, lohi, lo, hi, band=bandf * r, single=single)
            got = ctx.fmtstar_wavefront(r, L.GOAL_BALL, goal, band=bandf * r, single=single)
            assert got["status"] == ref["status"] and got["z"] - 1 == ref["z"] and got["cost"] == ref["cost"]
            assert got["collision_checks"] == ref["collision_checks"] and got["info"]["iters"] == ref["iters"]
            assert np.array_equal(got["A"] - 1, ref["A"]) and np.array_equal(got["C"], ref["C"]) and np.array_equal(got["path"] - 1, ref["path"])


def test_di_wavefront_matches_sequential_and_oracle(ctx, orc):
    """Kinodynamic FMT* (double integrator, BASELINE configs[3]) with the recursion on the device: the directed form (forward
    sets = rows of the cost matrix, transposed on the device).  One node per batch = mpfmt_di_fmtstar = orc_di_fmtstar exactly;
    with a band = the oracle's batched directed loop on the same graph, mask and segment counts."""
    from test_gpu_parity import di_world
    X, lohi, ss_lo, ss_hi = di_world(2500, 77)
    ctx.upload_samples(X)
    ctx.upload_boxes(lohi, ss_lo, ss_hi)
    oc, orow, oval, _ = orc.di_pairwise(X, 1.0, 1.0)
    for kind, goal, gd in ((L.GOAL_POINT, X[-1], 4), (L.GOAL_BALL, np.array([0.9, 0.9, 0.1]), 2)):
        seq = ctx.di_fmtstar(1.0, 1.0, kind, goal)
        ref = orc.di_fmtstar(X, 1.0, 1.0, oc, orow, oval, kind, goal, lohi, ss_lo, ss_hi)
        one = ctx.di_fmtstar_wavefront(1.0, 1.0, kind, goal, single=True)
        same_solution(one, seq)
        assert one["status"] == ref["status"] and one["collision_checks"] == ref["collision_checks"]
        assert np.array_equal(one["A"] - 1, ref["A"]) and np.array_equal(one["path"] - 1, ref["path"]) and np.array_equal(one["C"], ref["C"])
        ctx.di_graph(1.0, 1.0)                                  # (sets the wrapper's nnz; the resident graph is reused)
        mask, nseg = ctx.di_graph_edges_free()
        F = orc.pack(np.array([orc.is_free_state(x[:2], lohi) and bool(np.all((ss_lo <= x) & (x <= ss_hi))) for x in X]))
        for bandf in (0.0, 0.2, 1.0):
            want = orc.fmt_wavefront_directed(X, 2, oc, orow, oval, mask, nseg, F, kind, goal, band=bandf)
            got = ctx.di_fmtstar_wavefront(1.0, 1.0, kind, goal, band=bandf)
            assert got["status"] == want["status"] and got["z"] - 1 == want["z"] and got["cost"] == want["cost"]
            assert got["collision_checks"] == want["collision_checks"] and got["info"]["iters"] == want["iters"]
            assert np.array_equal(got["A"] - 1, want["A"]) and np.array_equal(got["C"], want["C"]) and np.array_equal(got["path"] - 1, want["path"])
            if seq["status"] == 1:
                assert got["status"] == 1 and got["cost"] <= seq["cost"] * 1.3
    # the Euclidean solve afterwards is unaffected by the directed state
    w = world(3000, 2, 20, 5)
    upload(ctx, w)
    a = ctx.fmtstar_wavefront(w.r, L.GOAL_BALL, w.goal_params(), single=True)
    b = ctx.fmtstar(w.r, L.GOAL_BALL, w.goal_params())
    same_solution(a, b)


@pytest.mark.parametrize("car", ["dubins", "reedsshepp"])
def test_car_wavefront_matches_sequential(ctx, car):
    """Dubins (directed) and Reeds-Shepp (structurally symmetric) planners with the recursion on the device: one node per batch
    equals the host recursion exactly (tree, costs, path, per-segment collision counts); a band still solves."""
    from test_gpu_parity import _car_world
    rng = np.random.default_rng(70 + 1500)
    X, lohi, lo, hi = _car_world(rng, 1500, 12)
    X[0] = [0.05, 0.05, 0.6]; X[-1] = [0.95, 0.95, 0.8]
    ctx.upload_samples(X)
    ctx.upload_boxes(lohi, lo, hi, dw=2)
    goal = np.array([0.95, 0.95, 0.08])
    rt, r = 0.05, (0.25 if car == "dubins" else 0.2)
    seq = getattr(ctx, car + "_fmtstar")(rt, 1.0, r, L.GOAL_BALL, goal)
    one = ctx.car_fmtstar_wavefront(car, rt, 1.0, r, L.GOAL_BALL, goal, single=True)
    same_solution(one, seq)
    band = ctx.car_fmtstar_wavefront(car, rt, 1.0, r, L.GOAL_BALL, goal, band=0.3 * r)
    assert band["status"] == seq["status"]
    if seq["status"] == 1:
        assert seq["cost"] * (1 - 1e-12) <= band["cost"] <= seq["cost"] * 1.3


@pytest.mark.parametrize("N,d,M,seed,bandf,single", [(3000, 2, 25, 5, 0.3, False), (3000, 2, 25, 5, 50.0, False), (20, 2, 3, 6, 0.0, True),
                                                      (6000, 3, 40, 7, 2.0, False)])
def test_run_path_counts_equal_the_stepwise_loop(ctx, N, d, M, seed, bandf, single):
    """mpfmt_wf_run enqueues steps in groups (8; 32 in single-node mode); the steps enqueued behind the goal batch must be
    void -- batch size, batch total and the batch list of the run path equal the one-step-at-a-time loop, also when the band
    takes the whole open set (|Z| > N / 8) and when N is smaller than a group (single mode, N < 32)."""
    w = world(N, d, M, seed, goal_radius=0.2)
    upload(ctx, w)
    ctx.wf_begin(w.r, L.GOAL_BALL, w.goal_params(), band=bandf * w.r, single=single)
    infos = []
    while True:
        info = ctx.wf_step()
        infos.append(info)
        if info["done"]:
            break
    zs_step = np.sort(ctx.wf_batch())
    res_step = ctx.wf_finish()
    got = ctx.fmtstar_wavefront(w.r, L.GOAL_BALL, w.goal_params(), band=bandf * w.r, single=single)
    zs_run = np.sort(ctx.wf_batch())
    gi, si = got["info"], infos[-1]
    assert gi["done"] == si["done"] and gi["iters"] == si["iters"]
    assert gi["nz"] == si["nz"] and gi["tot_z"] == si["tot_z"], (gi, si)
    assert gi["nx"] == si["nx"] and gi["tot_x"] == si["tot_x"] and gi["tot_conn"] == si["tot_conn"]
    assert np.array_equal(zs_run, zs_step) and len(zs_run) == gi["nz"]
    assert got["status"] == res_step["status"] and got["z"] == res_step["z"] and got["cost"] == res_step["cost"]
    assert got["collision_checks"] == res_step["collision_checks"]
    assert np.array_equal(got["A"], res_step["A"])
