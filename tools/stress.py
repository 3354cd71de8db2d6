"""Randomised parity stress: r-disc graph (both pair kernels, random shards) and graph sweep against the oracle on
random sizes / dimensions / radii / obstacle counts.  Usage: python tools/stress.py [seconds] [seed] [max d]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import motionplanning_jl_amd as mp
from motionplanning_jl_amd.distributed import DevArray
from oracle import oracle as orc
FORMS = {}                                                                       # steps by the form of their edge tests (include/mpfmt.h, stat sweep_form)
def run(budget=60.0, seed=0, maxd=8, max_cases=None):
    rng = np.random.default_rng(seed)
    t0 = time.time(); cases = 0; edges = 0
    while time.time() - t0 < budget and (max_cases is None or cases < max_cases):
        d = int(rng.integers(1, maxd + 1))
        N = int(rng.choice([1, 2, 17, 64, 65, 200, 700, 1500, 3000, 9000]))
        M = int(rng.choice([0, 1, 7, 60, 257, 520]))
        X = rng.random((N, d))
        if rng.random() < 0.3:
            X[rng.integers(0, N, N // 3)] = X[rng.integers(0, N)]              # duplicates
        if rng.random() < 0.3:
            X = 0.5 + 0.05 * (X - 0.5)                                          # a tight cluster
        c = rng.random((M, d)); h = 0.02 + 0.2 * rng.random((M, d)) * rng.random()
        lohi = np.stack([c - h, c + h], axis=1) if M else np.zeros((0, 2, d))
        inset = (rng.random() < 0.5)                                            # (a sample outside the state space keeps the edge tests out of the pair kernel)
        lo, hi = np.full(d, 0.05 * rng.random() * inset), np.full(d, 1 - 0.05 * rng.random() * inset)
        # radius for a target mean degree
        deg = float(rng.choice([0.5, 5, 40, 150, N]))
        span = X.max(0) - X.min(0) if N > 1 else np.ones(d)
        r = float(np.prod(np.maximum(span, 1e-3)) * deg / max(N, 1)) ** (1.0 / d) * 0.6
        oc, orow, oval = orc.rdisc_graph(X, r)
        world = int(rng.choice([1, 1, 2, 3, 5]))
        tot = 0
        for rank in range(world):
            ctx = mp.Context(0)
            if world > 1:
                ctx.set_shard(rank, world)
            ctx.upload_samples(X); ctx.upload_boxes(lohi, lo, hi)
            outs = []
            for path in (1, 2):
                ctx.set_option("rdisc_path", path)
                try:
                    outs.append(ctx.rdisc_graph(r))
                except mp.MPFMTError as e:
                    assert path == 2 and e.code == mp._lib.ERR_ARG
            ctx.set_option("rdisc_path", 0)
            for o in outs[1:]:
                for a, b in zip(outs[0], o):
                    assert np.array_equal(a, b), ("paths differ", d, N, r, world, rank)
            auto = ctx.rdisc_graph(r)                                  # automatic path choice; also leaves a valid graph in the ctx
            for a, b in zip(outs[0], auto):
                assert np.array_equal(a, b), ("auto path differs", d, N, r, world, rank)
            colptr, rowval, nzval = auto
            k = np.diff(colptr)
            if world == 1:
                assert np.array_equal(colptr - 1, oc) and np.array_equal(rowval - 1, orow), ("graph", d, N, r)
                assert np.array_equal(nzval, oval), ("costs", d, N, r)
            else:
                for v in np.flatnonzero(k):
                    assert np.array_equal(rowval[colptr[v] - 1:colptr[v + 1] - 1] - 1, orow[oc[v]:oc[v + 1]]), ("shard col", d, N, r, world, rank, v)
            tot += int(k.sum())
            m = ctx.graph_edges_free()
            want = orc.graph_edges_free(X, colptr - 1, rowval - 1, lohi, lo, hi)
            assert np.array_equal(m, want), ("sweep", d, N, M, r, world, rank)
            if len(rowval):
                cols = np.repeat(np.arange(1, N + 1), k)
                assert np.array_equal(ctx.edges_free(rowval, cols), want), ("edges_free", d, N, M, r)
            # the single-synchronisation step: first call careful, repeats speculative -- same resident graph and mask
            for rep in range(3):
                nnz = ctx.graph_step_device(r)
                assert nnz == len(rowval), ("step nnz", d, N, r, world, rank, rep)
                FORMS[ctx.stat("sweep_form")] = FORMS.get(ctx.stat("sweep_form"), 0) + 1
                cp, rv, nz, fr = ctx.graph_device_ptrs()
                dev = lambda ptr, n, ts: torch.as_tensor(DevArray(ptr, n, ts), device="cuda:0").cpu().numpy()
                assert np.array_equal(dev(cp, N + 1, "<i8"), colptr - 1), ("step colptr", d, N, r, world, rank, rep)
                if nnz:
                    assert np.array_equal(dev(rv, nnz, "<i4"), rowval - 1) and np.array_equal(dev(nz, nnz, "<f8"), nzval), ("step graph", d, N, r, world, rank, rep)
                    assert np.array_equal(dev(fr, (nnz + 63) // 64, "<i8").view(np.uint64), want), ("step mask", d, N, M, r, world, rank, rep)
            ctx.close()
        assert tot == len(orow), ("shard total", d, N, r, world)
        cases += 1; edges += len(orow)

    return cases, edges


if __name__ == "__main__":
    t0 = time.time()
    cases, edges = run(float(sys.argv[1]) if len(sys.argv) > 1 else 60.0, int(sys.argv[2]) if len(sys.argv) > 2 else 0,
                       int(sys.argv[3]) if len(sys.argv) > 3 else 8)
    print("stress ok: %d cases, %d edges, %.0f s; steps by edge-test form: %s" % (cases, edges, time.time() - t0, sorted(FORMS.items())))
