"""Randomised stress of the double-integrator build's matrix-core prefilter (kernels_di_mfma.hip): random workspaces (dimension 1 / 2, extents,
offsets far from the origin), velocity ranges, rho, radii from tiny to larger than the workspace, clustered / repeated / resting states.
For every world the graph of the vector-ALU path (di_path = 1) and of the automatic choice (matrix cores wherever the fp16 error bound
allows) must be the same arrays, and the NUMBER of pairs that reached the Newton iteration (stat `survivors`) must be equal -- the fp64
tests behind both filters are identical, so a pair the fp16 filter wrongly dropped shows there even when it would not have become an edge.
Small worlds are also compared with the oracle.   Usage: python tools/stress_di.py [seconds] [seed]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import motionplanning_jl_amd as mp
from oracle import oracle as orc


def run(budget=60.0, seed=0):
    rng = np.random.default_rng(seed)
    t0 = time.time(); cases = 0; mf_cases = 0; edges = 0; pairs = 0
    while time.time() - t0 < budget:
        m = int(rng.choice([1, 2, 2, 2]))
        N = int(rng.choice([70, 500, 3000, 9000, 20000, 50000]))
        scale = float(10.0 ** rng.uniform(-1, 1.5))
        offset = float(rng.choice([0.0, 0.0, 5.0, -40.0, 1000.0])) * scale
        vmax = float(10.0 ** rng.uniform(-1, 0.5)) * scale
        rho = float(10.0 ** rng.uniform(-0.7, 0.7))
        # radii around the interesting range: a typical optimal cost between neighbouring states is ~ (distance / vmax) scale
        r = float(10.0 ** rng.uniform(-0.7, 0.6)) * (scale / max(vmax, 1e-9)) * (N ** (-1.0 / (2 * m))) * 3.0
        X = np.concatenate([offset + scale * rng.random((N, m)), vmax * (2 * rng.random((N, m)) - 1)], axis=1)
        kind = rng.random()
        if kind < 0.2:
            X[: N // 3, :m] = offset + scale * (0.5 + 0.02 * rng.standard_normal((N // 3, m)))      # a cluster
        elif kind < 0.35:
            X[: N // 10, m:] = 0.0                                                                     # states at rest
        if N > 100 and rng.random() < 0.3:
            X[N // 2: N // 2 + 7] = X[:7]                                                              # repeated states
        got = {}
        try:
            for path in (1, 0):
                with mp.Context(0) as c:
                    c.set_option("di_path", path)
                    c.upload_samples(X)
                    got[path] = c.di_graph(rho, r) + (c.stat("survivors"), c.stat("di_path_used"))
        except mp.MPFMTError as e:
            if "capacity" in str(e).lower() or "memory" in str(e).lower():
                continue
            raise
        a, b = got[1], got[0]
        for u, v in zip(a[:4], b[:4]):
            assert np.array_equal(u, v), ("graph differs", m, N, scale, offset, vmax, rho, r)
        if b[5] == 2:
            assert a[4] == b[4], ("pairs reaching the iteration differ", a[4], b[4], m, N, scale, offset, vmax, rho, r)
            mf_cases += 1
        if N <= 3000:
            oc, orow, oval, otv = orc.di_pairwise(X, rho, r)
            assert np.array_equal(b[0] - 1, oc) and np.array_equal(b[1] - 1, orow) and np.array_equal(b[2], oval) and np.array_equal(b[3], otv), \
                ("oracle differs", m, N, scale, offset, vmax, rho, r)
        cases += 1; edges += len(b[1]); pairs += N * (N - 1)
    print("stress_di ok: %d cases (%d through the matrix cores), %d pairs, %d edges, %.0f s" % (cases, mf_cases, pairs, edges, time.time() - t0))


if __name__ == "__main__":
    run(float(sys.argv[1]) if len(sys.argv) > 1 else 60.0, int(sys.argv[2]) if len(sys.argv) > 2 else 0)
