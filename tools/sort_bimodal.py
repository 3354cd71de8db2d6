"""Is the run-to-run spread of the column-ordering kernel (0.92 / 1.05 / 1.14 ms with one binary on one box) tied to where the
arrays land?  One step loop per process, prints the kernel's time and the device addresses of the CSC arrays."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import motionplanning_jl_amd as mp
W = mp.workloads.north_star()
c = mp.Context(0); c.set_option("rebuild_index", 1)
c.upload_samples(W.X); c.upload_boxes(W.lohi, W.ss_lo, W.ss_hi)
for _ in range(3): c.graph_step_device(W.r)
c.timing_reset()
for _ in range(20): c.graph_step_device(W.r)
p = c.graph_device_ptrs()
print("sort %.3f pair %.3f sweep %.3f  ptrs %s" % (c.timing("rdisc_sort")[0], c.timing("rdisc_count")[0], c.timing("sweep_graph")[0],
                                                 " ".join("%x" % (x or 0) for x in p)))
