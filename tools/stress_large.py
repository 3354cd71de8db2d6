"""Randomised parity stress at sizes the half build shards into many tiles, slices and item regions (N 2e4 .. 4e5): the
single-synchronisation step with every (half build, edge-test form) setting against
  * the graph of scipy's cKDTree (an independent CPU code; pairs within r) and this library's non-MFMA pair kernel,
  * the oracle's sweep of that graph (orc.graph_edges_free), bit for bit.
The oracle's own r-disc graph takes minutes at these sizes, which is why tools/stress.py stops at N = 9000.
Usage: python tools/stress_large.py [seconds] [seed]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from scipy.spatial import cKDTree
import motionplanning_jl_amd as mp
from motionplanning_jl_amd.distributed import DevArray
from oracle import oracle as orc


def run(budget=60.0, seed=0, timed_form_only=False, max_pairs=1.2e8, sizes=(20000, 50000, 110000, 250000, 400000)):
    """timed_form_only: every case runs the DEFAULT options only (what bench.py times: half build + fused edge tests, form 2) over a
    cold step, a repeat and a re-upload -- the in-suite form (tests/test_gpu_step_parity.py), where most steps must be (1, 2)."""
    rng = np.random.default_rng(seed)
    t0 = time.time(); cases = 0; edges = 0; forms = {}
    dev = lambda ptr, n, ts: torch.as_tensor(DevArray(ptr, n, ts), device="cuda:0").cpu().numpy()
    while time.time() - t0 < budget:
        d = int(rng.integers(2, 7))
        N = int(rng.choice(list(sizes)))
        M = int(rng.choice([30, 200, 256] if timed_form_only else [0, 30, 200, 256, 400]))      # (form 2 takes <= 256 boxes)
        deg = float(rng.choice([6, 25, 60]))
        X = rng.random((N, d))
        kind = rng.random()
        if kind < 0.25:
            X[: N // 4] = 0.5 + (0.25 / 6.0) ** (1.0 / d) * (X[: N // 4] - 0.5)   # a clump six times as dense as the field
        elif kind < 0.4:
            X[:, 0] = X[:, 0] ** 3                                              # a density gradient
        r = float((deg / N) ** (1.0 / d) * 0.62)
        c = rng.random((M, d)); h = 0.02 + 0.15 * rng.random((M, d)) * rng.random()
        lohi = np.stack([c - h, c + h], axis=1) if M else np.zeros((0, 2, d))
        inset = 0.0 if timed_form_only else 0.03 * rng.random() * (rng.random() < 0.3)                     # (a sample outside the state space keeps the edge tests out of the pair kernel)
        lo, hi = np.full(d, inset), np.full(d, 1 - inset)
        tree = cKDTree(X)
        if tree.count_neighbors(tree, r) > max_pairs:
            continue                                                            # keeps one case under half a minute of host work
        pairs = tree.query_pairs(r, output_type="ndarray")
        col = np.concatenate([pairs[:, 0], pairs[:, 1]]); row = np.concatenate([pairs[:, 1], pairs[:, 0]])
        o = np.lexsort((row, col)); col, row = col[o], row[o]
        oc = np.concatenate([[0], np.cumsum(np.bincount(col, minlength=N))]).astype(np.int64)
        want = orc.graph_edges_free(X, oc, row.astype(np.int64), lohi, lo, hi)
        if os.environ.get("STRESS_DRY"):
            print("dry", d, N, M, deg, len(row), "%.0f s" % (time.time() - t0), flush=True); cases += 1; continue
        ctx = mp.Context(0)
        ctx.upload_samples(X); ctx.upload_boxes(lohi, lo, hi)
        ctx.set_option("rdisc_path", 1)
        colptr, rowval, nzval = ctx.rdisc_graph(r)
        ctx.set_option("rdisc_path", 0)
        assert np.array_equal(colptr - 1, oc) and np.array_equal(rowval - 1, row), ("kd-tree graph", d, N, r)
        dd = np.sqrt(((X[row] - X[col]) ** 2).sum(1))
        assert np.all(np.abs(nzval - dd) <= 1e-15 * dd + 1e-300), ("costs", d, N, r)
        for half in (1, 0):
            for form in (2, 1, 0):
                if timed_form_only and (half, form) != (1, 2):
                    continue
                if half == 0 and form != int(rng.integers(0, 3)):
                    continue                                                    # the whole build: one form per case
                ctx.set_option("rdisc_half", half); ctx.set_option("fuse_broad", form); ctx.set_option("rebuild_index", 1)
                for rep in range(3 if timed_form_only else 2):
                    ctx.upload_samples(X); ctx.upload_boxes(lohi, lo, hi)       # new samples: the step builds its graph again
                    nnz = ctx.graph_step_device(r)
                    assert nnz == len(rowval), ("step nnz", d, N, M, r, half, form, rep)
                    cp, rv, nz, fr = ctx.graph_device_ptrs()
                    assert np.array_equal(dev(cp, N + 1, "<i8"), oc), ("step colptr", d, N, M, r, half, form, rep)
                    assert np.array_equal(dev(rv, nnz, "<i4"), row), ("step rows", d, N, M, r, half, form, rep)
                    assert np.array_equal(dev(nz, nnz, "<f8"), nzval), ("step costs", d, N, M, r, half, form, rep)
                    assert np.array_equal(dev(fr, (nnz + 63) // 64, "<i8").view(np.uint64), want), ("step mask", d, N, M, r, half, form, rep)
                    key = (ctx.stat("rdisc_half_used"), ctx.stat("sweep_form"))
                    forms[key] = forms.get(key, 0) + 1
        ctx.close()
        cases += 1; edges += len(row)
        print("case %d: d %d N %d M %d edges %d, %.0f s" % (cases, d, N, M, len(row), time.time() - t0), flush=True)
    return cases, edges, forms


if __name__ == "__main__":
    t0 = time.time()
    cases, edges, forms = run(float(sys.argv[1]) if len(sys.argv) > 1 else 60.0, int(sys.argv[2]) if len(sys.argv) > 2 else 0)
    print("stress_large ok: %d cases, %d edges, %.0f s; steps by (half build used, edge-test form): %s"
          % (cases, edges, time.time() - t0, sorted(forms.items())))
