"""Generates the golden fixtures under tests/golden/ with the CPU oracle (oracle/mpfmt_oracle.c).

The reference ships no expected outputs (test/runtests.jl is a placeholder) and Julia is not
installed, so these vectors pin the build's declared canonical arithmetic, not reference output.
Every vector is cross-checked at generation time against the independent pure-Python
transliteration of the Julia lines (tests/jl_transliteration.py); a mismatch aborts.

Run from the repo root:  python tests/golden/make_golden.py
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from oracle import oracle as orc  # noqa: E402
import jl_transliteration as jl  # noqa: E402
import motionplanning_jl_amd as mp  # noqa: E402


def boxes(name):
    fx = json.load(open(os.path.join(HERE, "boxes_nd.json")))
    return np.array(fx[name], dtype=np.float64)          # (M, 2, d)


def degenerate_segments(lohi, rng):
    """axis-parallel segments, endpoints on faces / corners, zero-length, inside starts."""
    M, _, d = lohi.shape
    P, Q = [], []
    for k in range(M):
        lo, hi = lohi[k]
        c = 0.5 * (lo + hi)
        for i in range(d):
            a = c.copy(); b = c.copy()
            a[i] = lo[i] - 0.07; b[i] = hi[i] + 0.07          # axis-parallel through the centre
            P.append(a); Q.append(b)
            P.append(b); Q.append(a)
            a2 = c.copy(); b2 = c.copy()
            a2[i] = hi[i]; b2[i] = hi[i]                      # sliding along a closed face
            j = (i + 1) % d
            a2[j] = lo[j] - 0.05; b2[j] = hi[j] + 0.05
            P.append(a2); Q.append(b2)
        P.append(lo.copy()); Q.append(lo - 0.1)               # starts on the lo corner
        P.append(hi.copy()); Q.append(hi + 0.1)
        P.append(c.copy()); Q.append(c.copy())                # zero length inside
        P.append(c.copy()); Q.append(hi + 0.2)                # starts inside, exits
        P.append(lo - 0.2); Q.append(c.copy())                # ends inside
        P.append(lo - 0.1); Q.append(lo - 0.1)                # zero length outside
    return np.array(P), np.array(Q)


def gen_segments(name, seed):
    lohi = boxes(name)
    M, _, d = lohi.shape
    rng = np.random.default_rng(seed)
    P = rng.random((256, d)); Q = rng.random((256, d))
    # short segments too (FMT*-like edge lengths)
    P2 = rng.random((256, d)); Q2 = P2 + 0.12 * (rng.random((256, d)) - 0.5)
    Pd, Qd = degenerate_segments(lohi, rng)
    P = np.concatenate([P, P2, Pd]); Q = np.concatenate([Q, Q2, Qd])
    ss_lo, ss_hi = np.zeros(d), np.ones(d)
    free_boxes = np.array([orc.motion_free_boxes(p, q, lohi) for p, q in zip(P, Q)])
    free_full = np.array([orc.is_free_motion(p, q, lohi, ss_lo, ss_hi) for p, q in zip(P, Q)])
    pt_free = np.array([orc.is_free_state(p, lohi, ss_lo, ss_hi) for p in P])
    bl = [(list(b[0]), list(b[1])) for b in lohi]
    for e, (p, q) in enumerate(zip(P, Q)):
        assert jl.is_free_motion_boxes(p.tolist(), q.tolist(), bl) == free_boxes[e], (name, e)
        assert jl.is_free_motion(p.tolist(), q.tolist(), bl, list(ss_lo), list(ss_hi)) == free_full[e], (name, e)
        assert jl.is_free_state(p.tolist(), bl, list(ss_lo), list(ss_hi)) == pt_free[e], (name, e)
    np.savez_compressed(os.path.join(HERE, "segments_%s.npz" % name), P=P, Q=Q, lohi=lohi, ss_lo=ss_lo, ss_hi=ss_hi,
                        free_boxes=free_boxes, free_full=free_full, point_free=pt_free)
    print(name, "segments", len(P), "free", int(free_full.sum()))


def gen_rdisc(tag, N, d, seed):
    rng = np.random.default_rng(seed)
    X = rng.random((N, d))
    X[7] = X[3]                       # an exact duplicate pair (distance 0 neighbours)
    r = mp.workloads.fmt_radius(1.0, d, 1.0, N)
    colptr, rowval, nzval = orc.rdisc_graph(X, r, mode=0)
    # cross-check a few columns against the transliteration (1-based) and the KD-tree path
    V = X.tolist()
    kd = orc.KDTree(X)
    for v in (0, 3, 7, N // 2, N - 1):
        inds, ds = jl.inball_tree(V, v + 1, r)
        a, b = colptr[v], colptr[v + 1]
        assert [i - 1 for i in inds] == list(rowval[a:b]), (tag, v)
        assert np.array_equal(np.array(ds), nzval[a:b]), (tag, v)
        ki, kds = kd.inball(v, r)
        assert np.array_equal(ki, rowval[a:b]) and np.array_equal(kds, nzval[a:b]), (tag, v)
    np.savez_compressed(os.path.join(HERE, "rdisc_%s.npz" % tag), X=X, r=r, colptr=colptr, rowval=rowval.astype(np.int32),
                        nzval=nzval)
    print(tag, "nnz", len(rowval), "r", r)


def gen_fmt_cfg1():
    w = mp.workloads.cfg1()
    res = orc.fmtstar(w.X, w.r, orc.GOAL_BALL, w.goal_params(), w.lohi, w.ss_lo, w.ss_hi, init_idx=0, checkpts=True)
    bl = [(list(b[0]), list(b[1])) for b in w.lohi]
    V = w.X.tolist()
    gc, gr = list(w.goal_center), w.goal_radius
    ref = jl.fmtstar(V, w.r, lambda v: jl.is_goal_ball(v, gc, gr), bl, list(w.ss_lo), list(w.ss_hi), checkpts=True)
    assert ref["status"] == bool(res["status"]) and ref["cost"] == res["cost"], (ref["cost"], res["cost"])
    assert ref["collision_checks"] == res["collision_checks"]
    assert ref["path"] == [p + 1 for p in res["path"]]
    assert ref["A"] == [a + 1 for a in res["A"]]
    assert ref["C"] == list(res["C"])
    np.savez_compressed(os.path.join(HERE, "fmt_cfg1.npz"), X=w.X, lohi=w.lohi, r=w.r, goal=w.goal_params(),
                        A=res["A"], C=res["C"], path=res["path"], status=res["status"], cost=res["cost"],
                        collision_checks=res["collision_checks"], z=res["z"])
    print("fmt cfg1: status", res["status"], "cost", res["cost"], "checks", res["collision_checks"], "path", len(res["path"]))


if __name__ == "__main__":
    gen_segments("BOXES2D", 11)
    gen_segments("BOXES3D", 12)
    gen_rdisc("d2_n1000", 1000, 2, 21)
    gen_rdisc("d6_n1500", 1500, 6, 22)
    gen_fmt_cfg1()
