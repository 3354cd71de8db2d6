"""Golden fixtures for the 2-D SAT world: the reference's obstacle fixtures (test/obstaclesets/2D.jl, data
only), hand-derived known answers, and seeded random / degenerate segments whose masks come from the CPU oracle and are
cross-checked against the independent pure-Python transliteration (tests/jl_transliteration.py); a mismatch aborts.

Run from the repo root:  python tests/golden/make_golden_2d.py
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


def box(xr, yr):
    return ["polygon", [[xr[0], yr[0]], [xr[1], yr[0]], [xr[1], yr[1]], [xr[0], yr[1]]]]


# test/obstaclesets/2D.jl:3-41 (Box2D(xr, yr) = 4-point polygon, SAT2D.jl:59-62)
WORLDS = {
    "ISRR_2H": [box([.0, .16], [.36, .5]), box([.4, .5], [.19, .35]), box([.22, .46], [.57, .75]), box([.75, 1.], [.64, .77]),
                box([.22, .8], [.34, .51])],
    "TRI_BALLS": [["polygon", [[.3, .3], [.7, .3], [.5, .65]]], ["circle", [.3, .3], .15], ["circle", [.7, .3], .15],
                  ["circle", [.5, .65], .15]],
    "ISRR_POLY": [["polygon", [[.0, .25], [.27, .28], [.17, .4], [.0, .4]]],
                  ["polygon", [[.5, .2], [.2, .5], [.25, .7], [.4, .8], [.6, .8], [.7, .5]]],
                  ["polygon", [[.55, .2], [.75, .5], [.85, .5], [.85, .2]]], ["circle", [.9, .65], .1]],
    "ISRR_POLY_WITH_SPIKE": [["polygon", [[.0, .25], [.27, .28], [.17, .4], [.0, .4]]],
                             ["polygon", [[.5, .2], [.2, .5], [.25, .7], [.4, .8], [.6, .8], [.7, .5]]],
                             ["polygon", [[.55, .2], [.75, .5], [.85, .5], [.85, .2]]],
                             ["polygon", [[.3, .6], [.15, .85], [.4, .6]]], ["circle", [.9, .65], .1]],
    "EMPTY_2D": [],
}

# hand-derived (geometry you can check on paper): (world, v, w, is_free_motion, is_free_state(v))
KNOWN = [
    ["TRI_BALLS", [.1, .3], [.5, .3], False, True],       # through the centre of the circle at (.3,.3)
    ["TRI_BALLS", [.05, .05], [.1, .1], True, True],      # outside the compound's AABB
    ["TRI_BALLS", [.3, .3], [.3, .3], False, False],      # a point on a circle centre
    ["TRI_BALLS", [.45, .4], [.55, .4], False, True],     # inside the triangle, clear of the circles: the segment is
                                                          # caught by SAT, the endpoints are NOT (colliding(p, Polygon)
                                                          # is `@all [!ininterval(...)]` in the reference, SAT2D.jl:124-127)
    ["TRI_BALLS", [.1, .9], [.9, .9], True, True],        # above everything
    ["ISRR_2H", [.3, .25], [.6, .25], False, True],       # crosses Box2D([.4,.5],[.19,.35])
    ["ISRR_2H", [.3, .10], [.6, .10], True, True],        # passes below it
    ["ISRR_2H", [.45, .25], [.45, .27], False, True],     # inside that box (endpoints inside a polygon are "free states")
    ["ISRR_POLY", [.8, .65], [1.0, .65], False, False],   # starts inside the circle at (.9,.65)? no: |(.8,.65)-(.9,.65)| = .1 <= r
    ["EMPTY_2D", [.1, .1], [.9, .9], True, True],
]


def to_jl(shapes):
    parts = [jl.Circle(s[1], s[2]) if s[0] == "circle" else jl.Polygon(s[1]) for s in shapes]
    return jl.Compound2D(parts)


def to_orc(shapes):
    return orc.Shapes2D([("circle", tuple(s[1]), s[2]) if s[0] == "circle" else ("polygon", [tuple(p) for p in s[1]]) for s in shapes])


def segments(shapes, rng, n_random):
    P, Q = [rng.random((n_random, 2)) * 1.2 - 0.1], [rng.random((n_random, 2)) * 1.2 - 0.1]
    short = rng.random((n_random, 2))
    P.append(short); Q.append(short + 0.08 * (rng.random((n_random, 2)) - 0.5))
    deg_p, deg_q = [], []
    for s in shapes:
        if s[0] == "circle":
            c, r = np.array(s[1]), s[2]
            for a in np.linspace(0, 2 * np.pi, 8, endpoint=False):
                u = np.array([np.cos(a), np.sin(a)])
                deg_p += [c + r * u, c - 2 * r * u, c + r * u + r * perp(u), c, c + 0.5 * r * u]   # on the rim, through, tangent, centre, inside
                deg_q += [c + 2 * r * u, c + 2 * r * u, c + r * u - r * perp(u), c, c + 0.6 * r * u]
        else:
            pts = np.array(s[1]); n = len(pts); cen = pts.mean(0)
            for i in range(n):
                a, b = pts[i], pts[(i + 1) % n]
                deg_p += [a, a, 0.5 * (a + b), cen, cen, a - 0.05 * (b - a)]       # along an edge, vertex to vertex, edge midpoint out, inside
                deg_q += [b, a, 0.5 * (a + b) + 2 * (0.5 * (a + b) - cen), cen, a, b + 0.05 * (b - a)]
    if deg_p:
        P.append(np.array(deg_p)); Q.append(np.array(deg_q))
    return np.concatenate(P), np.concatenate(Q)


def perp(u):
    return np.array([u[1], -u[0]])


def main():
    out = {"worlds": WORLDS, "known": KNOWN}
    json.dump(out, open(os.path.join(HERE, "shapes_2d.json"), "w"), indent=1)
    ss_lo, ss_hi = np.zeros(2), np.ones(2)
    for w_i, (name, shapes) in enumerate(WORLDS.items()):
        J, O = to_jl(shapes), to_orc(shapes)
        for c in KNOWN:
            if c[0] != name:
                continue
            assert jl.is_free_motion_2d(c[1], c[2], J) == c[3], c
            assert jl.is_free_state_2d(c[1], J) == c[4], c
            assert bool(orc.unpack(orc.motions_free_2d([c[1]], [c[2]], O), 1)[0]) == c[3], c
            assert bool(orc.unpack(orc.points_free_2d([c[1]], O), 1)[0]) == c[4], c
        rng = np.random.default_rng(40 + w_i)
        P, Q = segments(shapes, rng, 400)
        for ss in ((None, None), (ss_lo, ss_hi)):
            mo = orc.unpack(orc.motions_free_2d(P, Q, O, *ss), len(P))
            po = orc.unpack(orc.points_free_2d(P, O, *ss), len(P))
            mj = np.array([jl.is_free_motion_2d(tuple(p), tuple(q), J, *ss) for p, q in zip(P, Q)])
            pj = np.array([jl.is_free_state_2d(tuple(p), J, *ss) for p in P])
            assert np.array_equal(mo, mj), (name, np.flatnonzero(mo != mj)[:5])
            assert np.array_equal(po, pj), (name, np.flatnonzero(po != pj)[:5])
        np.savez_compressed(os.path.join(HERE, "segments2d_%s.npz" % name), P=P, Q=Q, ss_lo=ss_lo, ss_hi=ss_hi,
                            free_motion=orc.unpack(orc.motions_free_2d(P, Q, O), len(P)),
                            free_motion_ss=orc.unpack(orc.motions_free_2d(P, Q, O, ss_lo, ss_hi), len(P)),
                            free_state=orc.unpack(orc.points_free_2d(P, O), len(P)),
                            free_state_ss=orc.unpack(orc.points_free_2d(P, O, ss_lo, ss_hi), len(P)))
        print(name, len(P), "segments; free", int(mo.sum()))


if __name__ == "__main__":
    main()
