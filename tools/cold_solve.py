"""Cold device solve on the north-star world, phase by phase (fresh ctx each time): upload (PCIe), step, solve -- with and without the
position-space copies.  usage: python tools/cold_solve.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import motionplanning_jl_amd as mp

def main():
    w = mp.workloads.north_star()
    for pos in ((1, 1) if len(sys.argv) > 1 else (1, 2, 0, 1, 2, 0)):
        c = mp.Context(0)
        c.set_option("wf_pos_space", pos)
        torch.cuda.synchronize()
        t0 = time.perf_counter(); c.upload_samples(w.X); c.upload_boxes(w.lohi, w.ss_lo, w.ss_hi); torch.cuda.synchronize()
        t1 = time.perf_counter()
        out = c.fmtstar_wavefront(w.r, mp._lib.GOAL_BALL, w.goal_params(), band=0.25 * w.r, want_tree=False)
        t2 = time.perf_counter()
        out2 = c.fmtstar_wavefront(w.r, mp._lib.GOAL_BALL, w.goal_params(), band=0.25 * w.r, want_tree=False)
        t3 = time.perf_counter()
        out3 = c.fmtstar_wavefront(w.r, mp._lib.GOAL_BALL, w.goal_params(), band=0.25 * w.r, want_tree=False)
        t4 = time.perf_counter()
        print(f"pos_space {pos}: upload {1e3*(t1-t0):.2f} ms  first solve {1e3*(t2-t1):.2f} ms  second {1e3*(t3-t2):.2f}  third {1e3*(t4-t3):.2f}  cost {out['cost']:.6f}", flush=True)
        c.close()

if __name__ == "__main__":
    main()
