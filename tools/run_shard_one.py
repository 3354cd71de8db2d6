import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import motionplanning_jl_amd as mp
g, G = int(sys.argv[1]), int(sys.argv[2])
w = mp.workloads.north_star()
c = mp.Context(0)
c.set_shard(g, G)
c.upload_samples(w.X); c.upload_boxes(w.lohi, w.ss_lo, w.ss_hi)
c.set_option("rebuild_index", 1)
for _ in range(12):
    nnz = c.graph_step_device(w.r)
torch.cuda.synchronize()
print("nnz", nnz, "slices", c.stat("slices"), "list_cap", c.stat("list_cap"))
