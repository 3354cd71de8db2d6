import sys, os, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
import motionplanning_jl_amd as mp
g, G, blocks, halo = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
w = mp.workloads.north_star()
c = mp.Context(0)
c.set_shard(g, G)
c.set_option("shard_blocks", blocks); c.set_option("index_halo", halo)
c.upload_samples(w.X); c.upload_boxes(w.lohi, w.ss_lo, w.ss_hi)
c.set_option("rebuild_index", 1)
for _ in range(12):
    nnz = c.graph_step_device(w.r)
torch.cuda.synchronize()
print("nnz", nnz)
