"""Experiment: do two contexts on ONE GPU, stepped through mpfmt_graph_step_launch / _finish from one thread, overlap?
(each context has its own stream; the kernels of a step are latency/issue bound, so a second stream might fill the gaps.)
Prints ms per step for one context alone and per step with two contexts in flight."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))

import motionplanning_jl_amd as mp

def main():
    W = mp.workloads.north_star()
    ctxs = []
    for k in range(2):
        c = mp.Context(0)
        c.set_option("rebuild_index", 1)
        c.upload_samples(W.X)
        c.upload_boxes(W.lohi, W.ss_lo, W.ss_hi)
        ctxs.append(c)
    for c in ctxs:
        for _ in range(3):
            c.graph_step_device(W.r)
    K = 20
    t0 = time.perf_counter()
    for _ in range(K):
        ctxs[0].graph_step_device(W.r)
    t1 = time.perf_counter()
    print("one ctx      : %.3f ms/step" % ((t1 - t0) / K * 1e3))
    t0 = time.perf_counter()
    for _ in range(K):
        for c in ctxs: c.graph_step_launch(W.r)
        for c in ctxs: c.graph_step_finish()
    t1 = time.perf_counter()
    print("two ctxs     : %.3f ms/step (%.3f ms per pair)" % ((t1 - t0) / (2 * K) * 1e3, (t1 - t0) / K * 1e3))
    # software pipeline: launch A; loop { launch B; finish A; launch A; finish B }
    t0 = time.perf_counter()
    ctxs[0].graph_step_launch(W.r)
    for _ in range(K):
        ctxs[1].graph_step_launch(W.r)
        ctxs[0].graph_step_finish()
        ctxs[0].graph_step_launch(W.r)
        ctxs[1].graph_step_finish()
    ctxs[0].graph_step_finish()
    t1 = time.perf_counter()
    print("pipelined    : %.3f ms/step" % ((t1 - t0) / (2 * K + 1) * 1e3))

main()
