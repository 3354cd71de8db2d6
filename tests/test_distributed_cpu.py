"""world_size-2 gloo test (CPU) of the N>1 exchange: per-shard free-edge masks are padded, all-gathered and
split back; the assembled shards equal the oracle's masks.  The oracle is the compute stand-in here
(tests only) -- the product path computes the shard masks on the GPU (bench.py --gpus N)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as tmp

import motionplanning_jl_amd as mp


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import oracle as orc
    w = mp.workloads.cfg1()
    N = w.N
    colptr, rowval, _ = orc.rdisc_graph(w.X, w.r)
    # shard = contiguous column range (stand-in for the library's cell-sorted tile range)
    a, b = N * rank // world, N * (rank + 1) // world
    if rank == 1:
        b = a                                  # ragged case: one shard is EMPTY
    cp = colptr[a:b + 1] - colptr[a]
    rv = rowval[colptr[a]:colptr[b]]
    full_cp = np.zeros(N + 1, dtype=np.int64)
    full_cp[a + 1:b + 1] = cp[1:]
    full_cp[b + 1:] = cp[-1] if len(cp) else 0
    local = orc.graph_edges_free(w.X, full_cp, rv, w.lohi, w.ss_lo, w.ss_hi) if len(rv) else np.zeros(0, np.uint64)
    t = torch.from_numpy(local.view(np.int64).copy())
    gathered, counts = mp.distributed.all_gather_mask(t, dist, world)
    parts = mp.distributed.split_gathered(gathered, counts)
    # the steady-state exchange (one collective per step): first call learns the lengths, the second runs on the agreed
    # capacity, the third makes rank 0's shard outgrow it (every rank must take the repeat-at-exact-size branch), the fourth
    # runs on the enlarged capacity again
    mg = mp.distributed.MaskGather(dist, world)
    seq = []
    for it in range(4):
        mine = t if it != 2 else (torch.cat([t, t, t]) if rank == 0 else t)
        g2, c2 = mg(mine)
        seq.append([p.numpy().copy() for p in mp.distributed.split_gathered(g2, c2)])
        want0 = t if it != 2 else torch.cat([t, t, t])
        if rank == 0:
            assert np.array_equal(seq[-1][0], want0.numpy())
    # shard sizes shrink, then grow again inside the same capacity: the rows must stay ZERO padded beyond counts[g]
    # (stale words of the longer previous mask used to stay in the padding)
    ones = torch.full((64,), -1, dtype=torch.int64)         # same length on every rank (rank 1's own shard is empty)
    for n in (len(ones), 3, len(ones) // 2, 0, 5):
        g3, c3 = mg(ones[:n] if rank == 0 else t)
        assert int(c3[0]) == n
        row = g3[0].numpy()
        assert (row[:n] == -1).all() and not row[n:].any(), (rank, n)
    q.put((rank, [p.numpy().view(np.uint64).copy() for p in parts], local, seq))
    dist.barrier()
    dist.destroy_process_group()


def test_all_gather_mask_world2_gloo():
    world = 2
    ctx = tmp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    res.sort(key=lambda x: x[0])
    locals_ = [r[2] for r in res]
    for rank, _, _, seq in res:                      # MaskGather: every step, every rank holds every shard's words
        for it, parts in enumerate(seq):
            for g in range(world):
                want = locals_[g].view(np.int64)
                if it == 2 and g == 0:
                    want = np.concatenate([want, want, want])
                assert np.array_equal(parts[g], want), (rank, it, g)
    for rank, parts, _, _ in res:
        assert len(parts) == world
        for g in range(world):
            assert np.array_equal(parts[g], locals_[g]), (rank, g)
    assert len(locals_[1]) == 0 and len(locals_[0]) > 0
