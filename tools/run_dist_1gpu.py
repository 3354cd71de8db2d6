#!/usr/bin/env python3
"""Exercise the multi-GPU step (mpfmt sharded_step + RCCL all_gather) with a 1-rank nccl group on one GPU."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
import numpy as np, torch, torch.distributed as dist
import motionplanning_jl_amd as mp
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
dev = torch.device("cuda", 0)
w = mp.workloads.cfg2(50000)
ctx = mp.Context(0)
ctx.set_stream(torch.cuda.current_stream(dev).cuda_stream)
ctx.set_shard(0, 1)
ctx.upload_samples(w.X); ctx.upload_boxes(w.lohi, w.ss_lo, w.ss_hi)
nnz, gathered, counts = mp.distributed.sharded_step(ctx, w.r, dist, 1, dev)
torch.cuda.synchronize()
ref = ctx.graph_edges_free()
got = gathered[0, :int(counts[0])].cpu().numpy().view(np.uint64)
assert np.array_equal(got, ref), "all-gathered mask differs"
gather = mp.distributed.MaskGather(dist, 1, dev)
for it in range(3):                                   # one collective per step from the second call on
    nnz2, g2, c2 = mp.distributed.sharded_step(ctx, w.r, dist, 1, dev, gather)
    torch.cuda.synchronize()
    assert nnz2 == nnz and np.array_equal(g2[0, :int(c2[0])].cpu().numpy().view(np.uint64), ref), "MaskGather step %d differs" % it
print("dist 1-rank ok: nnz", nnz, "words", int(counts[0]))
dist.destroy_process_group()
