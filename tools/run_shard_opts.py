"""One rank of a sharded north-star step under given options, for rocprofv3 (kernel times of a shard) or plain timing:
   python tools/run_shard_opts.py RANK WORLD [shard_blocks] [index_halo] [name=value ...]     (prints the step's wall time over 30 steps)"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import motionplanning_jl_amd as mp
g, G = int(sys.argv[1]), int(sys.argv[2])
blocks = int(sys.argv[3]) if len(sys.argv) > 3 else 1
halo = int(sys.argv[4]) if len(sys.argv) > 4 else 1
w = mp.workloads.north_star()
c = mp.Context(0)
c.set_shard(g, G)
c.set_option("shard_blocks", blocks); c.set_option("index_halo", halo)
for kv in sys.argv[5:]:
    k, v = kv.split("=")
    c.set_option(k, int(v))
c.upload_samples(w.X); c.upload_boxes(w.lohi, w.ss_lo, w.ss_hi)
c.set_option("rebuild_index", 1)
for _ in range(6):
    nnz = c.graph_step_device(w.r)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(30):
    nnz = c.graph_step_device(w.r)
torch.cuda.synchronize()
print("rank %d of %d %s: nnz %d step %.3f ms" % (g, G, " ".join(sys.argv[3:]), nnz, (time.perf_counter() - t0) / 30 * 1e3))
