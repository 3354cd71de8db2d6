import sys; sys.path.insert(0,'/root/repo')
import numpy as np, torch
import motionplanning_jl_amd as mp
w=mp.workloads.north_star()
c=mp.Context(0); c.set_option("rebuild_index",1); c.upload_samples(w.X); c.upload_boxes(w.lohi,w.ss_lo,w.ss_hi)
for k in range(5):
    X=mp.workloads.resample(w,k)
    t=torch.from_numpy(X).to("cuda:0"); torch.cuda.synchronize()
    c.upload_samples_device(t.data_ptr(), w.N, w.d)
    nnz=c.graph_step_device(w.r)
    print(k, nnz, "list_cap", c.stat("list_cap"), "list_max", c.stat("list_max"), "redo", c.stat("redo_count"), c.stat("redo_reason"), "qcap", c.stat("qcap"))
