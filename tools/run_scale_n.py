"""Single-GPU size sweep of the north-star workload family (R^6, M=200, r from the FMT* rule): robustness + scaling in N."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import motionplanning_jl_amd as mp
for N in (250_000, 1_000_000, 2_000_000, 4_000_000):
    w = mp.workloads.north_star(N)
    c = mp.Context(0)
    c.upload_samples(w.X); c.upload_boxes(w.lohi, w.ss_lo, w.ss_hi)
    c.set_option("rebuild_index", 1)
    nnz = c.graph_build_device(w.r); c.graph_sweep_device()
    c.timing_reset()
    t = time.time()
    for _ in range(3):
        nnz = c.graph_build_device(w.r); c.graph_sweep_device()
    c.graph_device_ptrs()
    import torch; torch.cuda.synchronize()
    dt = (time.time() - t) / 3
    km = {k: round(c.timing(k)[0], 3) for k in ("grid", "rdisc_count", "rdisc_fill", "rdisc_sort", "sweep_graph")}
    st = c.graph_stats() if hasattr(c, "graph_stats") else {}
    print("N %d r %.4f nnz %d deg %.1f step %.2f ms -> %.3g edges/s %.3g queries/s pool %d %s" % (N, w.r, nnz, nnz / N, dt * 1e3, nnz / dt, N / dt, c.stat("pool_used"), km), flush=True)
    c.close()
