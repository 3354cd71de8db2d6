"""Randomised parity stress of the rows beyond the hot path: 2-D SAT world, device sampler, Monte-Carlo edge counts,
Dubins steer.  Usage: python tools/stress2.py [seconds] [seed]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import motionplanning_jl_amd as mp
from oracle import oracle as orc
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
ctx = mp.Context(0)
t0 = time.time(); n_sat = n_smp = n_mc = n_dub = 0


def random_shapes(k):
    out = []
    while len(out) < k:
        c = rng.random(2)
        if rng.random() < 0.4:
            out.append(("circle", (float(c[0]), float(c[1])), float(0.01 + 0.15 * rng.random())))
        else:
            nv = int(rng.integers(3, 17))
            ang = np.sort(rng.random(nv) * 2 * np.pi)
            if np.max(np.diff(np.concatenate([ang, [ang[0] + 2 * np.pi]]))) > 0.9 * np.pi or np.min(np.diff(ang)) < 0.05:
                continue
            rad = 0.02 + 0.2 * rng.random()
            pts = [(float(c[0] + rad * np.cos(a)), float(c[1] + rad * np.sin(a))) for a in (ang if rng.random() < 0.5 else ang[::-1])]
            out.append(("polygon", pts))
    return out


while time.time() - t0 < budget:
    # --- 2-D SAT world: points, segments, graph ---
    shapes = random_shapes(int(rng.integers(0, 20)))
    S = orc.Shapes2D(shapes)
    lo, hi = np.array([0.02, 0.03]), np.array([0.97, 0.99])
    ss = (lo, hi) if rng.random() < 0.7 else (None, None)
    ctx.upload_shapes2d(shapes, *ss)
    n = int(rng.choice([1, 63, 64, 65, 500, 3000]))
    P = rng.random((n, 2)) * 1.2 - 0.1; Q = P + (rng.random((n, 2)) - 0.5) * float(rng.choice([0.02, 0.2, 1.0]))
    assert np.array_equal(ctx.states_free(P), orc.points_free_2d(P, S, *ss)), "sat2d points"
    assert np.array_equal(ctx.motions_free(P, Q), orc.motions_free_2d(P, Q, S, *ss)), "sat2d motions"
    X = rng.random((int(rng.choice([2, 200, 1500])), 2))
    ctx.upload_samples(X)
    colptr, rowval, _ = ctx.rdisc_graph(float(rng.choice([0.02, 0.08, 0.3])))
    assert np.array_equal(ctx.graph_edges_free(), orc.graph_edges_free_2d(X, colptr - 1, rowval - 1, S, *ss)), "sat2d graph"
    n_sat += 1
    # --- sampler in that world and in a box world ---
    N = int(rng.choice([1, 2, 100, 5000]))
    seed = int(rng.integers(0, 2**40))
    ctx.upload_shapes2d(shapes, lo, hi)
    Xs, att = ctx.sample_free(seed, N, init=[0.5, 0.5] if rng.random() < 0.5 else None)
    assert orc.unpack(orc.points_free_2d(Xs[1:] if len(Xs) > 1 else Xs[:0], S, lo, hi), max(len(Xs) - 1, 0)).all(), "sampler 2d"
    d = int(rng.integers(1, 9)); M = int(rng.choice([0, 5, 80]))
    c = rng.random((M, d)); h = 0.05 + 0.15 * rng.random((M, d))
    lohi = np.stack([c - h, c + h], axis=1) if M else np.zeros((0, 2, d))
    blo, bhi = np.full(d, -0.1), np.full(d, 1.2)
    ctx.upload_boxes(lohi, blo, bhi)
    gk = int(rng.integers(0, 3)); gc = rng.random(d)
    goal = {0: np.concatenate([gc - 0.1, gc + 0.1]), 1: np.concatenate([gc, [0.15]]), 2: gc}[gk]
    gct = int(rng.integers(0, 4))
    try:
        Xs, att = ctx.sample_free(seed, N, init=None, goal_kind=gk, goal_params=goal, goal_ct=gct)
    except mp.MPFMTError as e:
        assert e.code == mp._lib.ERR_INFEASIBLE, (str(e), N, d, M, gk)
        Xs = None
    if Xs is not None and N <= 100:                       # the scalar loop is the slow side
        rc, W, oatt = orc.sample_free(seed, N, d, None, lohi, blo, bhi, gk, goal, goal_ct=gct)
        assert rc == 0 and att == oatt and np.array_equal(Xs, W), ("sampler", N, d, M, gk, gct)
    n_smp += 1
    # --- Monte-Carlo counts ---
    Xm = rng.random((50, d))
    ctx.upload_samples(Xm); ctx.upload_boxes(lohi, blo, bhi)
    src = rng.integers(1, 51, 12); dst = rng.integers(1, 51, 12)
    sig = float(rng.choice([0.0, 0.01, 0.1])); R = int(rng.choice([1, 255, 256, 257, 2000]))
    assert np.array_equal(ctx.mc_edges_collision(src, dst, sig, R, seed=seed), orc.mc_edges(Xm, src - 1, dst - 1, sig, R, seed, lohi, blo, bhi)), "mc"
    n_mc += 1
    # --- Dubins steer ---
    A = np.column_stack([rng.random(300), rng.random(300), rng.random(300) * 2 * np.pi]); B = np.column_stack([rng.random(300), rng.random(300), rng.random(300) * 2 * np.pi])
    rt = float(rng.choice([0.02, 0.1, 0.5, 3.0]))
    cost, ctrl = ctx.dubins_steer(A, B, rt, 1.0)
    want = np.array([orc.dubins(a, b, rt, 1.0)[0] for a, b in zip(A, B)])
    assert np.array_equal(cost, want), ("dubins", np.max(np.abs(cost - want) / want))        # mp_math.h on both sides: bit-exact
    n_dub += 1
print("stress2 ok: sat2d %d, sampler %d, mc %d, dubins %d rounds, %.0f s" % (n_sat, n_smp, n_mc, n_dub, time.time() - t0))
