"""Worker of tests/test_gpu_multirank.py::test_one_thread_drives_all_ranks: ONE process, ONE thread, `world` ctxs on GPU 0 -- the
way a single Julia process would drive G GPUs -- with RCCL replaced by the tests' stand-in (MPFMT_RCCL_LIB), which
defers grouped collectives to ncclGroupEnd like RCCL does.  Per step: graph_step_launch on every ctx, graph_step_finish on every
ctx, then the mask gathers of all ctxs inside mpfmt_group_begin / _end; the NEXT step's kernels run before the gather is
finished (the overlap protocol), so the gathered masks must be the snapshot of the step they were launched for.  Checked
against an unsharded ctx: every shard's mask column by column, the lengths in the slot headers, the growth path (MPFMT_RETRY ->
relaunch in a group -> finish) and the shrink path.

usage: sp_group_worker.py WORLD [N] [d]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import motionplanning_jl_amd as mp  # noqa: E402
from motionplanning_jl_amd.distributed import DevArray  # noqa: E402

L = mp._lib
world = int(sys.argv[1])
N = int(sys.argv[2]) if len(sys.argv) > 2 else 30000
d = int(sys.argv[3]) if len(sys.argv) > 3 else 3
big = N >= 500000
w = mp.workloads.north_star(N) if big else mp.workloads.make("t", N, d, 40, 0.05, 0.12, seed=91, goal_radius=0.1)

from oracle import oracle as orc  # noqa: E402  (the checker)

radii = [w.r, w.r, w.r * (1.12 if big else 1.45), w.r * 0.8, w.r]        # steady, growth (x2 entries at the north star), shrink, back
want = {}
if big:
    # N = 1e6: the oracle cannot build the whole graph; the unsharded step's CSC + mask is the word-by-word reference of the shards,
    # and is itself compared with the ORACLE on sampled columns (KD-tree inball + is_free_motion of every entry of those columns)
    ref = mp.Context(0)
    ref.upload_samples(w.X); ref.upload_boxes(w.lohi, w.ss_lo, w.ss_hi)
    kd = orc.KDTree(w.X)
    rng = np.random.default_rng(17)
    for r in sorted(set(radii)):
        nnz = ref.graph_step_device(r)
        ptrs = ref.graph_device_ptrs()
        cp = torch.as_tensor(DevArray(ptrs[0], w.N + 1), device="cuda:0").cpu().numpy()
        mk = torch.as_tensor(DevArray(ptrs[3], (nnz + 63) // 64), device="cuda:0").cpu().numpy().view(np.uint64)
        rv = torch.as_tensor(DevArray(ptrs[1], nnz, "<i4"), device="cuda:0").cpu().numpy()
        bits = L.unpack_bits(mk, nnz)
        for v in rng.integers(0, w.N, size=300):
            oi, _ = kd.inball(int(v), r)
            a, b = int(cp[v]), int(cp[v + 1])
            assert np.array_equal(rv[a:b], oi), "unsharded reference column %d differs from the oracle" % v
            ob = orc.unpack(orc.edges_free(w.X, oi, np.full(len(oi), v), w.lohi, w.ss_lo, w.ss_hi), len(oi))
            assert np.array_equal(bits[a:b], ob), "unsharded reference mask of column %d differs from the oracle" % v
        want[r] = (cp.copy(), bits, nnz)
        del rv
    ref.close()                                              # (its logs at the largest radius are tens of GB: the shards need the room)
else:
    for r in sorted(set(radii)):
        oc, orow, _ = orc.rdisc_graph(w.X, r)
        want[r] = (oc, orc.unpack(orc.graph_edges_free(w.X, oc, orow, w.lohi, w.ss_lo, w.ss_hi), len(orow)), len(orow))

uid = L.comm_unique_id()
ctxs = [mp.Context(0) for _ in range(world)]
L.group_begin()
for g, c in enumerate(ctxs):
    c.comm_create(g, world, uid)
L.group_end()
for c in ctxs:
    c.set_option("rebuild_index", 1)
    c.upload_samples(w.X); c.upload_boxes(w.lohi, w.ss_lo, w.ss_hi)


def step_all(r):
    """one step on every ctx; returns the shards' nnz and their colptr (taken NOW: the next step overwrites the resident graph
    before this step's gather is checked)"""
    for c in ctxs:
        c.graph_step_launch(r)
    nnzs = [c.graph_step_finish() for c in ctxs]
    cps = [torch.as_tensor(DevArray(c.graph_device_ptrs()[0], w.N + 1), device="cuda:0").cpu().numpy() for c in ctxs]
    return nnzs, cps


def check(r, nnz_at_launch, cps, res):
    """every ctx holds every shard's mask; shard q's words against the reference mask of exactly q's columns, WORD BY WORD"""
    cp_ref, bits_ref, nnz_ref = want[r]
    deg_ref = np.diff(cp_ref)
    G0 = None
    owned = np.zeros(w.N, dtype=np.int32)
    for g, c in enumerate(ctxs):
        ptr, stride, words, nnzs = res[g]
        assert list(nnzs) == nnz_at_launch, (list(nnzs), nnz_at_launch)
        assert int(nnzs.sum()) == nnz_ref, (int(nnzs.sum()), nnz_ref)
        G = torch.as_tensor(DevArray(ptr, stride * world), device="cuda:0").cpu().numpy().view(np.uint64).reshape(world, stride)
        for q in range(world):
            assert G[q, 0] == words[q] == (nnzs[q] + 63) // 64 and G[q, 1] == nnzs[q]
        if g == 0:
            G0 = G.copy()
            for q in range(world):
                deg = np.diff(cps[q])
                own = np.flatnonzero(deg)
                owned[own] += 1
                assert np.array_equal(deg[own], deg_ref[own]), "shard %d: a column it owns is incomplete" % q
                ln = deg_ref[own]
                idx = np.repeat(cp_ref[own] - (np.cumsum(ln) - ln), ln) + np.arange(int(ln.sum()))
                exp = L.pack_bits(bits_ref[idx])[:int(words[q])]
                assert int(ln.sum()) == nnzs[q] and np.array_equal(G[q, 2:2 + words[q]], exp), "shard %d: mask words differ from the reference" % q
            assert np.all(owned[deg_ref > 0] == 1), "every column with neighbours belongs to exactly one shard"
        else:
            for q in range(world):
                assert np.array_equal(G[q, :2 + words[q]], G0[q, :2 + words[q]]), "ctx %d holds a different copy of shard %d" % (g, q)
    return int(bits_ref.sum())


retries = 0
nnz_prev, cps_prev = step_all(radii[0])
for k in range(len(radii)):
    r = radii[k]
    first = k == 0
    hint = (max(nnz_prev) + 63) // 64 + 64 if first else 0            # a single-thread driver has seen every shard: it can agree the first capacity itself
    L.group_begin()
    for c in ctxs:
        c.allgather_free_mask_launch(hint)
    L.group_end()
    nnz_launch, cps_launch = list(nnz_prev), cps_prev
    if k + 1 < len(radii):
        nnz_prev, cps_prev = step_all(radii[k + 1])                             # the next step overwrites every ctx's mask before the gather is finished
    res = [c.allgather_free_mask_finish(world, allow_retry=True) for c in ctxs]
    if any(x is None for x in res):
        assert all(x is None for x in res), "every ctx must see the same lengths and ask for the repeat"
        retries += 1
        L.group_begin()
        for c in ctxs:
            c.allgather_free_mask_relaunch()
        L.group_end()
        res = [c.allgather_free_mask_finish(world) for c in ctxs]
    check(r, nnz_launch, cps_launch, res)
assert retries >= 1, "the growth step must have gone through MPFMT_RETRY"
for c in ctxs:
    c.close()
print("group ok: world %d, N %d, %d retries" % (world, w.N, retries))
