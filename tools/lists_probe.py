"""Chunk-list lengths of the north-star tiles (on the GPU box): quantiles, the longest list and where it sits in the tile order."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import motionplanning_jl_amd as mp
w = mp.workloads.north_star()
c = mp.Context(0); c.set_option("rebuild_index", 1); c.upload_samples(w.X); c.upload_boxes(w.lohi, w.ss_lo, w.ss_hi)
for k in range(5):
    X = mp.workloads.resample(w, k)
    t = torch.from_numpy(X).to("cuda:0"); torch.cuda.synchronize()
    c.upload_samples_device(t.data_ptr(), w.N, w.d)
    nnz = c.graph_step_device(w.r)
    nt = c.graph_stats()["tiles"]
    print(k, nnz, "tiles", nt, "list_cap", c.stat("list_cap"), "sum", c.stat("list_sum"),
          "quantiles 0/100/500/900/990/999/1000 permille:", [c.stat("list_q%d" % q) for q in (0, 100, 500, 900, 990, 999, 1000)],
          "argmax tile", c.stat("list_argmax"), "redo", c.stat("redo_count"), c.stat("redo_reason"), "qcap", c.stat("qcap"))
