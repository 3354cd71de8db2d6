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

ref = mp.Context(0)
ref.upload_samples(w.X); ref.upload_boxes(w.lohi, w.ss_lo, w.ss_hi)
radii = [w.r, w.r, w.r * (1.12 if big else 1.45), w.r * 0.8, w.r]        # steady, growth (x2 entries at the north star), shrink, back
want = {}
for r in sorted(set(radii)):
    if big:
        nnz = ref.graph_step_device(r)
        ptrs = ref.graph_device_ptrs()
        cp = torch.as_tensor(DevArray(ptrs[0], w.N + 1), device="cuda:0").cpu().numpy()
        mk = torch.as_tensor(DevArray(ptrs[3], (nnz + 63) // 64), device="cuda:0").cpu().numpy().view(np.uint64)
        want[r] = (cp.copy(), mk.copy(), nnz)
    else:
        colptr, rowval, nzval = ref.rdisc_graph(r)
        want[r] = (colptr - 1, ref.graph_edges_free(), len(rowval))

ref.close()                                                  # (its logs at the largest radius are tens of GB: the shards need the room)
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
    for c in ctxs:
        c.graph_step_launch(r)
    return [c.graph_step_finish() for c in ctxs]


def check(r, nnz_at_launch, res):
    cp_ref, mask_ref, nnz_ref = want[r]
    bits_ref = L.unpack_bits(mask_ref, nnz_ref)
    tot = 0
    for g, c in enumerate(ctxs):
        ptr, stride, words, nnzs = res[g]
        assert list(nnzs) == nnz_at_launch, (list(nnzs), nnz_at_launch)
        assert int(nnzs.sum()) == nnz_ref, (int(nnzs.sum()), nnz_ref)
        G = torch.as_tensor(DevArray(ptr, stride * world), device="cuda:0").cpu().numpy().view(np.uint64).reshape(world, stride)
        for q in range(world):
            assert G[q, 0] == words[q] == (nnzs[q] + 63) // 64 and G[q, 1] == nnzs[q]
        if g == 0:
            tot = sum(int(L.unpack_bits(G[q, 2:2 + words[q]], nnzs[q]).sum()) for q in range(world))
            assert tot == int(bits_ref.sum()), "free edges of all shards != unsharded"
    return tot


retries = 0
nnz_prev = step_all(radii[0])
for k in range(len(radii)):
    r = radii[k]
    first = k == 0
    hint = (max(nnz_prev) + 63) // 64 + 64 if first else 0            # a single-thread driver has seen every shard: it can agree the first capacity itself
    L.group_begin()
    for c in ctxs:
        c.allgather_free_mask_launch(hint)
    L.group_end()
    nnz_launch = list(nnz_prev)
    if k + 1 < len(radii):
        nnz_prev = step_all(radii[k + 1])                             # the next step overwrites every ctx's mask before the gather is finished
    res = [c.allgather_free_mask_finish(world, allow_retry=True) for c in ctxs]
    if any(x is None for x in res):
        assert all(x is None for x in res), "every ctx must see the same lengths and ask for the repeat"
        retries += 1
        L.group_begin()
        for c in ctxs:
            c.allgather_free_mask_relaunch()
        L.group_end()
        res = [c.allgather_free_mask_finish(world) for c in ctxs]
    check(r, nnz_launch, res)
assert retries >= 1, "the growth step must have gone through MPFMT_RETRY"
for c in ctxs:
    c.close()
print("group ok: world %d, N %d, %d retries" % (world, w.N, retries))
