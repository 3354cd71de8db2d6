"""Worker of tests/test_gpu_multirank.py: one of WORLD_SIZE ranks that all use GPU 0 and exchange through the library's C ABI
(mpfmt_comm_*), RCCL replaced by the shared-memory stand-in named in MPFMT_RCCL_LIB.  Checks, against an unsharded ctx on the
same GPU: (1) the gathered free-edge masks of all shards are the unsharded mask, column by column; (2) the sharded wavefront
solve with one all-gather of (x, y_min, c_min) triples per wavefront gives the unsharded tree, checks and wavefront count."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import motionplanning_jl_amd as mp  # noqa: E402
from motionplanning_jl_amd.distributed import DevArray  # noqa: E402

L = mp._lib
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
w = mp.workloads.make("t", 40000, 3, 40, 0.05, 0.12, seed=77, goal_radius=0.1)
uid = torch.zeros(L.COMM_ID_BYTES, dtype=torch.uint8)
if rank == 0:
    uid.copy_(torch.frombuffer(bytearray(L.comm_unique_id()), dtype=torch.uint8))
dist.broadcast(uid, src=0)

ref = mp.Context(0)                       # the unsharded answer, computed by every rank for itself
ref.upload_samples(w.X); ref.upload_boxes(w.lohi, w.ss_lo, w.ss_hi)
colptr, rowval, nzval = ref.rdisc_graph(w.r)
refmask = L.unpack_bits(ref.graph_edges_free(), len(rowval))

c = mp.Context(0)
c.comm_create(rank, world, uid.numpy().tobytes())
c.set_option("rebuild_index", 1)
c.upload_samples(w.X); c.upload_boxes(w.lohi, w.ss_lo, w.ss_hi)
for step in range(4):                     # careful, speculative, then a radius change (capacity growth), and back
    r = w.r * (1.45 if step == 2 else 1.0)
    nnz = c.graph_step_device(r)
    c.allgather_free_mask_launch()
    ptr, stride, words, nnzs = c.allgather_free_mask_finish(world)
    assert words[rank] == (nnz + 63) // 64 and nnzs[rank] == nnz and int(nnzs.sum()) > 0
    if step != 2:
        assert int(nnzs.sum()) == len(rowval), (int(nnzs.sum()), len(rowval))
        G = torch.as_tensor(DevArray(ptr, stride * world), device="cuda:0").cpu().numpy().view(np.uint64).reshape(world, stride)
        # this rank's own shard against the unsharded mask: columns it owns, in CSC order
        cp = torch.as_tensor(DevArray(c.graph_device_ptrs()[0], w.N + 1), device="cuda:0").cpu().numpy()
        own = np.flatnonzero(np.diff(cp))
        idx = np.concatenate([np.arange(colptr[v] - 1, colptr[v + 1] - 1) for v in own])
        mine = L.unpack_bits(G[rank, 2:2 + words[rank]], nnz)
        assert np.array_equal(mine, refmask[idx]), "shard mask differs from the unsharded one"
        assert G[rank, 0] == words[rank] and G[rank, 1] == nnz
        # every rank holds every shard: the popcounts add up to the unsharded number of free edges
        tot = sum(int(L.unpack_bits(G[g, 2:2 + words[g]], nnzs[g]).sum()) for g in range(world))
        assert tot == int(refmask.sum())

want = ref.fmtstar_wavefront(w.r, L.GOAL_BALL, w.goal_params(), band=0.3 * w.r)
c.graph_step_device(w.r)
got = c.fmtstar_wavefront(w.r, L.GOAL_BALL, w.goal_params(), band=0.3 * w.r)       # sharded: one exchange per wavefront
assert got["status"] == want["status"] and got["z"] == want["z"] and got["cost"] == want["cost"]
assert np.array_equal(got["A"], want["A"]) and np.array_equal(got["C"], want["C"]) and np.array_equal(got["path"], want["path"])
assert got["collision_checks"] == want["collision_checks"], (got["collision_checks"], want["collision_checks"])
assert got["info"]["iters"] == want["info"]["iters"]
c.close(); ref.close()
dist.barrier()
if rank == 0:
    print("multirank ok: world %d, nnz %d, %d wavefronts, cost %.6f" % (world, len(rowval), got["info"]["iters"], got["cost"]))
dist.destroy_process_group()
