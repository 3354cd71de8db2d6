"""Which form of the edge tests a step ends up with over a grid of worlds (unit-cube state space, default options): prints
(pair kernel, half build used, edge-test form, pending-list overflow, builds redone so far, why) for a cold step, a repeat and a
step on new samples of each.  tests/test_gpu_step_parity.py::test_form_grid asserts the forms.
Usage: python tools/run_form_grid.py"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import motionplanning_jl_amd as mp



def grid(dims=(2, 3, 4, 6), sizes=(20000, 110000, 400000), boxes=(30, 256), degs=(6, 60), seed=5):
    """yields (d, N, M, deg, nnz, [(path, half, form, overflow, redone, reason) per step], vector-ALU filter used, kernel ms of the steps)"""
    rng = np.random.default_rng(seed)
    for d in dims:
        for N in sizes:
            for M in boxes:
                for deg in degs:
                    X = rng.random((N, d)); r = float((deg / N) ** (1.0 / d) * 0.62)
                    X2 = rng.random((N, d))
                    lohi = mp.workloads.make_boxes(rng, M, d, 0.05, 0.25, [])
                    with mp.Context(0) as c:
                        c.set_option("rebuild_index", 1)
                        out = []
                        for it in range(3):
                            c.upload_samples(X if it < 2 else X2); c.upload_boxes(lohi, np.zeros(d), np.ones(d))
                            nnz = c.graph_step_device(r)
                            out.append((c.stat("rdisc_path_used"), c.stat("rdisc_half_used"), c.stat("sweep_form"), c.stat("pend_overflowed"),
                                        c.stat("redo_count"), c.stat("redo_reason")))
                            flt = c.stat("filter_valu")
                            ms = c.timing("grid")[0] + c.timing("rdisc_count")[0] + c.timing("rdisc_sort")[0] + c.timing("sweep_graph")[0]
                    yield d, N, M, deg, nnz, out, flt, ms


if __name__ == "__main__":
    for d, N, M, deg, nnz, out, flt, ms in grid():
        print("d %d N %6d M %3d deg %2d nnz %9d: (path, half, form, overflow, redone, why) %s  filter %s  %.3f ms (timers, 3 steps)"
              % (d, N, M, deg, nnz, out, "fp64 VALU" if flt else "fp16 MFMA", ms), flush=True)
