"""Independent pure-Python restatement of the reference's hot-path Julia functions.

Purpose: a second, structurally different statement of the same reference lines that the C
oracle (oracle/mpfmt_oracle.c) follows, so that the oracle is cross-checked by something
other than itself (the reference ships no tests and Julia is not installed).  Python floats
are IEEE binary64 and CPython never fuses a*b+c, so arithmetic matches the declared canon.
Pure-Python loops: small cases only.  1-based indices are used where the reference uses them.
"""
import heapq
import math


# src/utilities/utils.jl:3-39 -- @any / @all are short-circuit loops over a comprehension
def jl_any(gen):
    for x in gen:
        if x:
            return True
    return False


def jl_all(gen):
    for x in gen:
        if not x:
            return False
    return True


# src/utilities/utils.jl:41-51
def blend(b, a1, a2):
    return [a1[j] if b[j] else a2[j] for j in range(len(b))]


# src/collisioncheckers/boxesND.jl:42
def is_free_state_box(v, lo, hi):
    return jl_any(not (lo[i] <= v[i] <= hi[i]) for i in range(len(lo)))


# src/collisioncheckers/boxesND.jl:43
def is_free_state_boxes(v, boxes):
    return jl_all(is_free_state_box(v, lo, hi) for (lo, hi) in boxes)


# src/collisioncheckers/boxesND.jl:44-45
def is_free_motion_broadphase(l, h, lo, hi):
    return jl_any(hi[i] < l[i] or lo[i] > h[i] for i in range(len(lo)))


def _div(a, b):
    """IEEE division (Python raises on /0)."""
    if b == 0.0:
        if a == 0.0 or a != a:
            return float("nan")
        neg = (math.copysign(1.0, a) < 0) != (math.copysign(1.0, b) < 0)
        return -math.inf if neg else math.inf
    return a / b


# src/collisioncheckers/boxesND.jl:46-51
def is_free_motion_box(v, w, lo, hi):
    n = len(lo)
    v_to_w = [w[i] - v[i] for i in range(n)]
    corner = blend([v[i] < lo[i] for i in range(n)], lo, hi)
    lambdas = [_div(corner[i] - v[i], v_to_w[i]) for i in range(n)]
    return not jl_any(
        jl_all(i == j or (lo[j] <= v[j] + v_to_w[j] * lambdas[i] <= hi[j]) for j in range(n))
        for i in range(n))


# src/collisioncheckers/boxesND.jl:52-56
def is_free_motion_boxes(v, w, boxes):
    bb_min = [min(a, b) for a, b in zip(v, w)]
    bb_max = [max(a, b) for a, b in zip(v, w)]
    return jl_all(is_free_motion_broadphase(bb_min, bb_max, lo, hi) or is_free_motion_box(v, w, lo, hi)
                  for (lo, hi) in boxes)


# src/statespaces.jl:150
def in_state_space(v, ss_lo, ss_hi):
    return jl_all(ss_lo[i] <= v[i] <= ss_hi[i] for i in range(len(v)))


# src/statespaces.jl:151-152 (s2w = Identity)
def is_free_state(v, boxes, ss_lo, ss_hi):
    return in_state_space(v, ss_lo, ss_hi) and is_free_state_boxes(v, boxes)


# src/collisioncheckers/boxesND.jl:26: is_free_motion(v, w, CC::PointRobotNDBoxes) = (CC.count += 1; is_free_motion(v, w, CC.boxes))
# CC = a dict holding `boxes` and the mutable `count` of boxesND.jl:15-23
def is_free_motion_cc(v, w, CC):
    CC["count"] += 1
    return is_free_motion_boxes(v, w, CC["boxes"])


# src/statespaces.jl:153-158 with collision_waypoints = (v, w) (geometric.jl:20).  The checker -- and with it the count
# increment of boxesND.jl:26 -- is only reached when in_state_space(wps[i]) held (&& short circuit, :155-156).
def is_free_motion(v, w, boxes, ss_lo, ss_hi, CC=None):
    wps = (v, w)
    if CC is None:
        CC = {"boxes": boxes, "count": 0}
    return jl_all(in_state_space(wps[i], ss_lo, ss_hi) and is_free_motion_cc(wps[i], wps[i + 1], CC)
                  for i in range(len(wps) - 1))


def sqeuclid(a, b):
    s = 0.0
    for i in range(len(a)):
        t = a[i] - b[i]
        s = t * t if i == 0 else s + t * t
    return s


# src/nearneighbors.jl:179-183 (tree semantics: reduced distance <= r^2), 1-based indices
def inball_tree(V, v, r):
    q = V[v - 1]
    inds = [i for i in range(1, len(V) + 1) if sqeuclid(q, V[i - 1]) <= r * r]   # inrange(..., sorted)
    if v in inds:
        inds.remove(v)                                                            # deleteat!(self)
    ds = [math.sqrt(sqeuclid(q, V[i - 1])) for i in inds]                         # colwise
    return inds, ds


# src/nearneighbors.jl:138-150 (generic fallback: sqrt distance <= r)
def inball_generic(V, v, r):
    allds = [math.sqrt(sqeuclid(V[v - 1], w)) for w in V]
    inds, ds = [], []
    for i in range(1, len(V) + 1):
        if i != v and allds[i - 1] <= r:
            inds.append(i)
            ds.append(allds[i - 1])
    return inds, ds


# src/nearneighbors.jl:104-107
def filter_neighborhood(n, f):
    inds, ds = n
    keep = [f[i] for i in inds]
    return [i for i, k in zip(inds, keep) if k], [d for d, k in zip(ds, keep) if k]


def is_goal_ball(v, center, radius):
    return math.sqrt(sqeuclid(v, center)) <= radius


# src/planners/fmt.jl:3-119, connections = :R, 1-based like the reference.
# V: list of points, V[0] is sample 1 (= init).  is_goal: predicate on a point.
def fmtstar(V, r, is_goal, boxes, ss_lo, ss_hi, checkpts=True, init_idx=1):
    N = len(V)
    CC = {"boxes": boxes, "count": 0}           # P.CC.count = 0, fmt.jl:12
    if not is_free_state(V[init_idx - 1], boxes, ss_lo, ss_hi):
        return None
    F = {i: True for i in range(1, N + 1)}
    if checkpts:
        for i in range(1, N + 1):
            F[i] = is_free_state(V[i - 1], boxes, ss_lo, ss_hi)
    A = {i: 0 for i in range(1, N + 1)}
    W = {i: True for i in range(1, N + 1)}
    H = {i: False for i in range(1, N + 1)}
    Cc = {i: 0.0 for i in range(1, N + 1)}
    cache = {}

    def near(v):
        if v not in cache:
            cache[v] = inball_tree(V, v, r)
        return cache[v]

    W[init_idx] = False
    H[init_idx] = True
    heap = [(0.0, init_idx)]
    z = heapq.heappop(heap)[1]
    while not is_goal(V[z - 1]):
        H_new = []
        xs, _ = filter_neighborhood(near(z), W)
        for x in xs:
            if checkpts and not F[x]:
                continue
            inds, ds = filter_neighborhood(near(x), H)
            costs = [Cc[y] + dd for y, dd in zip(inds, ds)]
            c_min = min(costs)
            y_idx = costs.index(c_min)          # findmin: first minimal element
            y_min = inds[y_idx]
            if is_free_motion(V[y_min - 1], V[x - 1], boxes, ss_lo, ss_hi, CC):      # counts inside the checker (boxesND.jl:26)
                A[x] = y_min
                Cc[x] = c_min
                heapq.heappush(heap, (c_min, x))
                H_new.append(x)
                W[x] = False
        for x in H_new:
            H[x] = True
        H[z] = False
        if heap:
            z = heapq.heappop(heap)[1]
        else:
            break
    sol = [z]
    while sol[0] != 1:
        sol.insert(0, A[sol[0]])
        if sol[0] == 0:
            break
    return dict(status=is_goal(V[z - 1]), cost=Cc[z], z=z, collision_checks=CC["count"],
                A=[A[i] for i in range(1, N + 1)], C=[Cc[i] for i in range(1, N + 1)], path=sol)


# ---- 2-D SAT world: src/collisioncheckers/SAT2D.jl, robots2D.jl, utilities/vec2Dutils.jl ----------------------------
# Shapes are plain dicts; every function below is the Julia line it cites, evaluated with Python floats (IEEE fp64,
# one rounding per operation, like Julia's un-fused scalar code).
import math as _m


def dot(a, b):                 # dot(::SVector{2}, ::SVector{2})
    return a[0] * b[0] + a[1] * b[1]


def cross(a, b):               # vec2Dutils.jl:7
    return a[0] * b[1] - a[1] * b[0]


def norm2(v):                  # vec2Dutils.jl:5
    return dot(v, v)


def perp(v):                   # vec2Dutils.jl:6
    return (v[1], -v[0])


def minmaxV(x, y):             # vec2Dutils.jl:35
    return (x, y) if x < y else (y, x)


def overlapping(i1, i2):       # vec2Dutils.jl:33
    return i1[0] <= i2[1] and i2[0] <= i1[1]


def ininterval(x, i):          # vec2Dutils.jl:34
    return i[0] <= x <= i[1]


def wrap1(i, n):               # vec2Dutils.jl:36 (1-based)
    return n if i < 1 else 1 if i > n else i


def projectNextrema(pts, n):   # vec2Dutils.jl:19-28
    dmin, dmax = _m.inf, -_m.inf
    for p in pts:
        d = dot(p, n)
        if d < dmin:
            dmin = d
        if d > dmax:
            dmax = d
    return (dmin, dmax)


def Circle(c, r):              # SAT2D.jl:14-28
    if r <= 0:
        raise ValueError("Radius must be positive")
    return dict(kind="circle", c=(c[0], c[1]), r=r, xrange=(c[0] - r, c[0] + r), yrange=(c[1] - r, c[1] + r))


def Polygon(points):           # SAT2D.jl:40-58
    pts = [(float(p[0]), float(p[1])) for p in points]
    N = len(pts)
    if N < 3:
        raise ValueError("Polygons need at least 3 points")
    s = 0.0
    for i in range(1, N + 1):
        a, b = pts[wrap1(i + 1, N) - 1], pts[i - 1]
        t = (a[0] - b[0]) * (a[1] + b[1])
        s = t if i == 1 else s + t
    if s > 0:
        pts.reverse()
    edges = [(pts[wrap1(i + 1, N) - 1][0] - pts[i - 1][0], pts[wrap1(i + 1, N) - 1][1] - pts[i - 1][1]) for i in range(1, N + 1)]
    normals = []
    for g in edges:
        p = perp(g)
        nrm = _m.sqrt(p[0] * p[0] + p[1] * p[1])
        normals.append((p[0] / nrm, p[1] / nrm))
    ang = [_m.atan2(n[1], n[0]) for n in normals] + [_m.atan2(normals[0][1], normals[0][0])]
    if any(-_m.pi <= ang[i + 1] - ang[i] <= 0 for i in range(N)):
        raise ValueError("Polygon must be convex")
    xs = [p[0] for p in pts]; ys = [p[1] for p in pts]
    return dict(kind="polygon", points=pts, edges=edges, normals=normals, xrange=(min(xs), max(xs)), yrange=(min(ys), max(ys)),
                nextrema=[projectNextrema(pts, n) for n in normals])


def Box2D(xr, yr):             # SAT2D.jl:59-62
    return Polygon([(xr[0], yr[0]), (xr[1], yr[0]), (xr[1], yr[1]), (xr[0], yr[1])])


def Compound2D(parts):         # SAT2D.jl:83-97
    if not parts:
        return dict(kind="compound", parts=[], xrange=(0.0, 0.0), yrange=(0.0, 0.0))
    return dict(kind="compound", parts=list(parts),
                xrange=(min(P["xrange"][0] for P in parts), max(P["xrange"][1] for P in parts)),
                yrange=(min(P["yrange"][0] for P in parts), max(P["yrange"][1] for P in parts)))


def Line(v, w):                # SAT2D.jl:66-81
    edge = (w[0] - v[0], w[1] - v[1])
    normal = perp(edge)
    return dict(kind="line", v=(v[0], v[1]), w=(w[0], w[1]), edge=edge, normal=normal, xrange=minmaxV(v[0], w[0]),
                yrange=minmaxV(v[1], w[1]), ndotv=dot(v, normal))


def AABBseparated(S1, S2):     # SAT2D.jl:119
    return not (overlapping(S1["xrange"], S2["xrange"]) and overlapping(S1["yrange"], S2["yrange"]))


def pointinAABB(p, S):         # SAT2D.jl:120
    return ininterval(p[0], S["xrange"]) and ininterval(p[1], S["yrange"])


def colliding_point(p, S):     # SAT2D.jl:121-133
    if S["kind"] == "circle":
        t = (p[0] - S["c"][0], p[1] - S["c"][1])
        return norm2(t) <= S["r"] ** 2
    if S["kind"] == "polygon":
        if not pointinAABB(p, S):
            return False
        return jl_all(not ininterval(dot(p, S["normals"][i]), S["nextrema"][i]) for i in range(len(S["normals"])))
    if not pointinAABB(p, S):
        return False
    return jl_any(colliding_point(p, P) for P in S["parts"])


def colliding_ends_free(L, S):  # SAT2D.jl:163-174,179-182
    if S["kind"] == "circle":
        if AABBseparated(L, S):
            return False
        vc = (S["c"][0] - L["v"][0], S["c"][1] - L["v"][1])
        d2 = norm2(L["edge"])
        if d2 * S["r"] ** 2 < cross(L["edge"], vc) ** 2:
            return False
        return 0 <= dot(vc, L["edge"]) <= d2
    if AABBseparated(L, S):
        return False
    if not ininterval(L["ndotv"], projectNextrema(S["points"], L["normal"])):          # is_separating_axis(L, P), :113
        return False
    return not jl_any(not overlapping(S["nextrema"][i], minmaxV(dot(L["v"], S["normals"][i]), dot(L["w"], S["normals"][i])))
                      for i in range(len(S["normals"])))                               # :112, :80


def colliding_line(L, S):      # SAT2D.jl:154-157,176-178
    if S["kind"] == "compound":
        if AABBseparated(S, L):
            return False
        return jl_any(colliding_line(L, P) for P in S["parts"])
    return colliding_ends_free(L, S) or colliding_point(L["v"], S) or colliding_point(L["w"], S)


def is_free_state_2d(v, obstacles, ss_lo=None, ss_hi=None):       # statespaces.jl:151-152 + robots2D.jl:12
    return (ss_lo is None or in_state_space(v, ss_lo, ss_hi)) and not colliding_point(v, obstacles)


def is_free_motion_2d(v, w, obstacles, ss_lo=None, ss_hi=None):   # statespaces.jl:153-158 + robots2D.jl:13-14
    return (ss_lo is None or in_state_space(v, ss_lo, ss_hi)) and not colliding_line(Line(v, w), obstacles)


# ---- Dubins car: src/statespaces/simplecars.jl:51-215, src/utilities/utils.jl:91 ------------------------------------------
TWOPI = 2 * _m.pi


def mod2piF(x):                # mod(x, 2pi) with Julia's float mod
    r = _m.fmod(x, TWOPI)
    if r == 0:
        return 0.0
    return r + TWOPI if r < 0 else r


def carsegment2stepcontrol(t, d):          # :91  StepControl(abs(d), (sign(d), t))
    return [abs(d), float((d > 0) - (d < 0)), float(t)]


def _dubins_words(d, a, b):
    """the six words in the order dubins() tries them (:206-211); yields (cnew, path) or None"""
    ca, sa, cb, sb = _m.cos(a), _m.sin(a), _m.cos(b), _m.sin(b)
    out = []
    tmp = 2 + d * d - 2 * (ca * cb + sa * sb - d * (sa - sb))                       # LSL :106
    if not tmp < 0:
        th = _m.atan2(cb - ca, d + sa - sb)
        t = mod2piF(-a + th); p = _m.sqrt(max(tmp, 0.0)); q = mod2piF(b - th)
        out.append((t + p + q, [(1, t), (0, p), (1, q)]))
    tmp = 2 + d * d - 2 * (ca * cb + sa * sb - d * (sb - sa))                       # RSR :121
    if not tmp < 0:
        th = _m.atan2(ca - cb, d - sa + sb)
        t = mod2piF(a - th); p = _m.sqrt(max(tmp, 0.0)); q = mod2piF(-b + th)
        out.append((t + p + q, [(-1, t), (0, p), (-1, q)]))
    tmp = d * d - 2 + 2 * (ca * cb + sa * sb - d * (sa + sb))                       # RSL :136
    if not tmp < 0:
        p = _m.sqrt(max(tmp, 0.0))
        th = _m.atan2(ca + cb, d - sa - sb) - _m.atan2(2.0, p)
        t = mod2piF(a - th); q = mod2piF(b - th)
        out.append((t + p + q, [(-1, t), (0, p), (1, q)]))
    tmp = -2 + d * d + 2 * (ca * cb + sa * sb + d * (sa + sb))                      # LSR :151
    if not tmp < 0:
        p = _m.sqrt(max(tmp, 0.0))
        th = _m.atan2(-ca - cb, d + sa + sb) - _m.atan2(-2.0, p)
        t = mod2piF(-a + th); q = mod2piF(-b + th)
        out.append((t + p + q, [(1, t), (0, p), (-1, q)]))
    tmp = (6 - d * d + 2 * (ca * cb + sa * sb + d * (sa - sb))) / 8                 # RLR :166
    if not abs(tmp) >= 1:
        p = TWOPI - _m.acos(tmp)
        th = _m.atan2(ca - cb, d - sa + sb)
        t = mod2piF(a - th + p / 2); q = mod2piF(a - b - t + p)
        out.append((t + p + q, [(-1, t), (1, p), (-1, q)]))
    tmp = (6 - d * d + 2 * (ca * cb + sa * sb - d * (sa - sb))) / 8                 # LRL :181
    if not abs(tmp) >= 1:
        p = TWOPI - _m.acos(tmp)
        th = _m.atan2(-ca + cb, d + sa - sb)
        t = mod2piF(-a + th + p / 2); q = mod2piF(b - a - t + p)
        out.append((t + p + q, [(1, t), (-1, p), (1, q)]))
    return out


def dubins(s1, s2, r=1.0, s=1.0):          # :198-215
    v = ((s2[0] - s1[0]) / r, (s2[1] - s1[1]) / r)
    d = _m.sqrt(v[0] * v[0] + v[1] * v[1])
    th = _m.atan2(v[1], v[0])
    a = mod2piF(s1[2] - th); b = mod2piF(s2[2] - th)
    cmin, pmin = _m.inf, [[0.0, 0.0, 0.0]] * 3
    for cnew, segs in _dubins_words(d, a, b):
        if not cmin <= cnew:
            cmin = cnew
            pmin = [carsegment2stepcontrol(t, dd) for (t, dd) in segs]
    pmin = [[u[0] * r, u[1], u[2] / r] for u in pmin]                              # scaleradius :92
    pmin = [[u[0] / s, u[1] * s, u[2]] for u in pmin]                              # scalespeed :93
    return cmin * r, pmin


def car_propagate(v, u):                   # :52-65
    t, s, invr = u
    if abs(t * s * invr) > 10 * 2.220446049250313e-16:
        return (v[0] + (_m.sin(v[2] + t * s * invr) - _m.sin(v[2])) / invr,
                v[1] + (_m.cos(v[2]) - _m.cos(v[2] + t * s * invr)) / invr, mod2piF(v[2] + t * s * invr))
    return (v[0] + t * s * _m.cos(v[2]), v[1] + t * s * _m.sin(v[2]), mod2piF(v[2] + t * s * invr))


def car_collision_waypoints(v, w, r=1.0, s=1.0):    # :68-83 + statespaces.jl:127-135
    _, us = dubins(v, w, r, s)
    path = []
    thres = _m.pi / 12
    v = tuple(v)
    for u in us:
        t, sp, invr = u
        m = _m.floor(t * sp * invr / thres)
        path.append(v)
        if m != 0:
            for i in range(1, m + 1):
                path.append((v[0] + (_m.sin(v[2] + i * thres) - _m.sin(v[2])) / invr,
                             v[1] + (_m.cos(v[2]) - _m.cos(v[2] + i * thres)) / invr, mod2piF(v[2] + i * thres)))
        v = car_propagate(v, u)
    path.append(tuple(w))
    return path


def car_is_free_motion(v, w, r, s, boxes, ss_lo, ss_hi):    # statespaces.jl:153-158; returns (free, segment tests made)
    wps = car_collision_waypoints(v, w, r, s)
    cnt = 0
    for i in range(len(wps) - 1):
        if not in_state_space(wps[i], ss_lo, ss_hi):
            return False, cnt
        cnt += 1
        if not is_free_motion_boxes(wps[i][:2], wps[i + 1][:2], boxes):
            return False, cnt
    return True, cnt


# ---- Reeds-Shepp car: src/statespaces/simplecars.jl:228-524 ------------------------------------------------------------------
def _R(x, y):                              # :230
    return _m.sqrt(x * x + y * y), _m.atan2(y, x)


def _M(t):                                 # :232-235
    m = mod2piF(t)
    return m - TWOPI if m > _m.pi else m


def _Tau(u, v, E, N):                      # :236-243
    delta = _M(u - v)
    A = _m.sin(u) - _m.sin(delta)
    B = _m.cos(u) - _m.cos(delta) - 1
    r, th = _R(E * A + N * B, N * A - E * B)
    t = 2 * _m.cos(delta) - 2 * _m.cos(v) - 2 * _m.cos(u) + 3
    return _M(th + _m.pi) if t < 0 else _M(th)


def _Omega(u, v, E, N, t):                 # :244
    return _M(_Tau(u, v, E, N) - u + v - t)


def _timeflip(s):                          # :245
    return (-s[0], s[1], -s[2])


def _reflect(s):                           # :246
    return (s[0], -s[1], -s[2])


def _backwards(s):                         # :247
    return (s[0] * _m.cos(s[2]) + s[1] * _m.sin(s[2]), s[0] * _m.sin(s[2]) - s[1] * _m.cos(s[2]), s[2])


def _LpSpLp(T, c):                         # :365-376
    tx, ty, tt = T
    r, th = _R(tx - _m.sin(tt), ty - 1 + _m.cos(tt))
    u = r; t = mod2piF(th); v = mod2piF(tt - t)
    cnew = t + u + v
    if c <= cnew:
        return None
    return cnew, [(1, t), (0, u), (1, v)]


def _LpSpRp(T, c):                         # :378-391
    tx, ty, tt = T
    r, th = _R(tx + _m.sin(tt), ty - 1 - _m.cos(tt))
    if r * r < 4:
        return None
    u = _m.sqrt(r * r - 4)
    r1, th1 = _R(u, 2.0)
    t = mod2piF(th + th1); v = mod2piF(t - tt)
    cnew = t + u + v
    if c <= cnew:
        return None
    return cnew, [(1, t), (0, u), (-1, v)]


def _LpRmLp(T, c):                         # :393-408
    tx, ty, tt = T
    E = tx - _m.sin(tt); N = ty + _m.cos(tt) - 1
    if E * E + N * N > 16:
        return None
    r, th = _R(E, N)
    u = _m.acos(1 - r * r / 8)
    t = mod2piF(th - u / 2 + _m.pi); v = mod2piF(_m.pi - u / 2 - th + tt)
    u = -u
    cnew = t - u + v
    if c <= cnew:
        return None
    return cnew, [(1, t), (-1, u), (1, v)]


def _LpRmLm(T, c):                         # :410-425
    tx, ty, tt = T
    E = tx - _m.sin(tt); N = ty + _m.cos(tt) - 1
    if E * E + N * N > 16:
        return None
    r, th = _R(E, N)
    u = _m.acos(1 - r * r / 8)
    t = mod2piF(th - u / 2 + _m.pi); v = mod2piF(_m.pi - u / 2 - th + tt) - TWOPI
    u = -u
    cnew = t - u - v
    if c <= cnew:
        return None
    return cnew, [(1, t), (-1, u), (1, v)]


def _LpRpuLmuRm(T, c):                     # :427-442
    tx, ty, tt = T
    E = tx + _m.sin(tt); N = ty - _m.cos(tt) - 1
    p = (2 + _m.sqrt(E * E + N * N)) / 4
    if p < 0 or p > 1:
        return None
    u = _m.acos(p)
    t = mod2piF(_Tau(u, -u, E, N)); v = mod2piF(_Omega(u, -u, E, N, tt)) - TWOPI
    cnew = t + 2 * u - v
    if c <= cnew:
        return None
    return cnew, [(1, t), (-1, u), (1, -u), (-1, v)]


def _LpRmuLmuRp(T, c):                     # :444-459
    tx, ty, tt = T
    E = tx + _m.sin(tt); N = ty - _m.cos(tt) - 1
    p = (20 - E * E - N * N) / 16
    if p < 0 or p > 1:
        return None
    u = -_m.acos(p)
    t = mod2piF(_Tau(u, u, E, N)); v = mod2piF(_Omega(u, u, E, N, tt))
    cnew = t - 2 * u + v
    if c <= cnew:
        return None
    return cnew, [(1, t), (-1, u), (1, u), (-1, v)]


def _LpRmSmLm(T, c):                       # :461-479
    tx, ty, tt = T
    E = tx - _m.sin(tt); N = ty + _m.cos(tt) - 1
    D, be = _R(E, N)
    if D < 2:
        return None
    ga = _m.acos(2 / D); F = _m.sqrt(D * D / 4 - 1)
    t = mod2piF(_m.pi + be - ga); u = 2 - 2 * F
    if u > 0:
        return None
    v = mod2piF(-3 * _m.pi / 2 + ga + tt - be) - TWOPI
    cnew = t + _m.pi / 2 - u - v
    if c <= cnew:
        return None
    return cnew, [(1, t), (-1, -_m.pi / 2), (0, u), (1, v)]


def _LpRmSmRm(T, c):                       # :481-497
    tx, ty, tt = T
    E = tx + _m.sin(tt); N = ty - _m.cos(tt) - 1
    D, be = _R(E, N)
    if D < 2:
        return None
    t = mod2piF(be + _m.pi / 2); u = 2 - D
    if u > 0:
        return None
    v = mod2piF(-_m.pi - tt + be) - TWOPI
    cnew = t + _m.pi / 2 - u - v
    if c <= cnew:
        return None
    return cnew, [(1, t), (-1, -_m.pi / 2), (0, u), (-1, v)]


def _LpRmSmLmRp(T, c):                     # :499-518
    tx, ty, tt = T
    E = tx + _m.sin(tt); N = ty - _m.cos(tt) - 1
    D, be = _R(E, N)
    if D < 2:
        return None
    ga = _m.acos(2 / D); F = _m.sqrt(D * D / 4 - 1)
    t = mod2piF(_m.pi + be - ga); u = 4 - 2 * F
    if u > 0:
        return None
    v = mod2piF(_m.pi + be - tt - ga)
    cnew = t + _m.pi - u + v
    if c <= cnew:
        return None
    return cnew, [(1, t), (-1, -_m.pi / 2), (0, u), (1, -_m.pi / 2), (-1, v)]


def reedsshepp(s1, s2, r=1.0, s=1.0):      # :265-363
    dx, dy = (s2[0] - s1[0]) / r, (s2[1] - s1[1]) / r
    ct, st = _m.cos(s1[2]), _m.sin(s1[2])
    target = (dx * ct + dy * st, -dx * st + dy * ct, mod2piF(s2[2] - s1[2]))
    tT = _timeflip(target); rT = _reflect(target); trT = _reflect(tT)
    bT = _backwards(target); btT = _timeflip(bT); brT = _reflect(bT); btrT = _reflect(btT)
    POST, POST_T, POST_R, POST_B, POST_R_T, POST_B_T, POST_B_R, POST_B_R_T = range(8)
    four = [(target, POST), (tT, POST_T), (rT, POST_R), (trT, POST_R_T)]
    eight = four + [(bT, POST_B), (btT, POST_B_T), (brT, POST_B_R), (btrT, POST_B_R_T)]
    c, p, post = _m.inf, None, None
    plan = [(_LpSpLp, four), (_LpSpRp, four), (_LpRmLp, [(target, POST), (rT, POST_R)]), (_LpRmLm, eight), (_LpRpuLmuRm, four),
            (_LpRmuLmuRp, four), (_LpRmSmLm, eight), (_LpRmSmRm, eight), (_LpRmSmLmRp, four)]
    for f, targets in plan:
        for T, code in targets:
            out = f(T, c)
            if out is not None:
                c, segs = out
                p = [carsegment2stepcontrol(t, d) for (t, d) in segs]
                post = code
    u = [[q[0] * r, q[1], q[2] / r] for q in p]
    u = [[q[0] / s, q[1] * s, q[2]] for q in u]
    if post in (POST_T, POST_R_T, POST_B_T, POST_B_R_T):
        u = [[q[0], -q[1], q[2]] for q in u]            # timeflip! :248-253
    if post in (POST_R, POST_R_T, POST_B_R, POST_B_R_T):
        u = [[q[0], q[1], -q[2]] for q in u]            # reflect! :254-259
    if post in (POST_B, POST_B_T, POST_B_R, POST_B_R_T):
        u = u[::-1]                                     # backwards! :260
    return c * r, u


def car_collision_waypoints_rs(v, w, r=1.0, s=1.0):     # :68-83 + statespaces.jl:127-135 with the Reeds-Shepp controls
    _, us = reedsshepp(v, w, r, s)
    path = []
    thres = _m.pi / 12
    v = tuple(v)
    for u in us:
        t, sp, invr = u
        m = _m.floor(t * sp * invr / thres)
        path.append(v)
        if m != 0:
            for i in range(1, m + 1):
                path.append((v[0] + (_m.sin(v[2] + i * thres) - _m.sin(v[2])) / invr,
                             v[1] + (_m.cos(v[2]) - _m.cos(v[2] + i * thres)) / invr, mod2piF(v[2] + i * thres)))
        v = car_propagate(v, u)
    path.append(tuple(w))
    return path


# ======================================================================================================================
# closest / closeR (boxesND.jl:61-86 through bvls.jl:19-218; SAT2D.jl:208-285).  numpy's LAPACK plays the part of Julia's
# (`\` = least squares, chol, eigfact), so this path shares no arithmetic with the C restatement's hand-written QR.
# ======================================================================================================================
import numpy as _np


def bvls(A, b, l, u):                       # bvls.jl:19-218; returns x, or None when the 10n iterations run out
    A = _np.asarray(A, dtype=float); b = _np.asarray(b, dtype=float)
    m, n = A.shape
    oopslist = _np.zeros(n, dtype=bool)
    state = _np.zeros(n, dtype=int)
    x = _np.zeros(n)
    atbound = _np.zeros(n, dtype=bool)
    between = _np.zeros(n, dtype=bool)
    criti = -1
    crits = 0
    myeps = 1.0e-10
    for i in range(n):
        if u[i] >= _np.inf and l[i] <= -_np.inf:
            x[i] = 0; state[i] = 0; between[i] = True
        elif u[i] >= _np.inf:
            x[i] = l[i]; state[i] = 1; atbound[i] = True
        elif l[i] <= -_np.inf:
            x[i] = u[i]; state[i] = 2; atbound[i] = True
        elif abs(l[i]) <= abs(u[i]):
            x[i] = l[i]; state[i] = 1; atbound[i] = True
        else:
            x[i] = u[i]; state[i] = 2; atbound[i] = True
    it = 0
    while it < 10 * n:
        it += 1
        grad = A.T @ (A @ x - b)
        grad[oopslist] = 0.0
        done = True
        for i in range(n):
            if (abs(grad[i]) > (1 + _np.linalg.norm(b)) * myeps and state[i] == 0) or (grad[i] < 0 and state[i] == 1) or \
                    (grad[i] > 0 and state[i] == 2):
                done = False
                break
        if done:
            return x
        newi = -1
        newg = 0.0
        for i in range(n):
            if atbound[i]:
                if i == criti:
                    continue
                if grad[i] > 0 and state[i] == 2 and abs(grad[i]) > newg:
                    newi = i; newg = abs(grad[i])
                if grad[i] < 0 and state[i] == 1 and abs(grad[i]) > newg:
                    newi = i; newg = abs(grad[i])
        if newi != -1:
            atbound[newi] = False; state[newi] = 0; between[newi] = True
        Aproj = A[:, between]
        An = A[:, atbound]
        bproj = b - An @ x[atbound] if atbound.any() else b
        z = _np.linalg.lstsq(Aproj, bproj, rcond=None)[0] if between.any() else _np.zeros(0)
        xnew = x.copy()
        xnew[between] = z
        if newi != -1 and ((xnew[newi] <= l[newi] and x[newi] == l[newi]) or (xnew[newi] >= u[newi] and x[newi] == u[newi])):
            oopslist[newi] = True
            if xnew[newi] <= l[newi] and state[newi] == 1:
                state[newi] = 1; x[newi] = l[newi]
            if xnew[newi] >= u[newi] and state[newi] == 2:
                state[newi] = 2; x[newi] = u[newi]
            atbound[newi] = True
            between[newi] = False
            continue
        oopslist[:] = False
        alpha = 1.0
        for i in range(n):
            if between[i]:
                if xnew[i] > u[i]:
                    newalpha = min(alpha, (u[i] - x[i]) / (xnew[i] - x[i]))
                    if newalpha < alpha:
                        criti = i; crits = 2; alpha = newalpha
                if xnew[i] < l[i]:
                    newalpha = min(alpha, (l[i] - x[i]) / (xnew[i] - x[i]))
                    if newalpha < alpha:
                        criti = i; crits = 1; alpha = newalpha
        x = x + alpha * (xnew - x)
        if alpha < 1:
            between[criti] = False; atbound[criti] = True; state[criti] = crits
        for i in range(n):
            if x[i] >= u[i]:
                x[i] = u[i]; state[i] = 2; between[i] = False; atbound[i] = True
            if x[i] <= l[i]:
                x[i] = l[i]; state[i] = 1; between[i] = False; atbound[i] = True
    return None


def closest_box(p, lo, hi, W):              # boxesND.jl:61-70
    p = _np.asarray(p, dtype=float)
    L = _np.linalg.cholesky(W).T            # chol(W): upper
    vmin = bvls(L, L @ p, lo, hi)
    if vmin is None:
        return None                         # the reference throws here (nothing - p)
    return float((vmin - p) @ (W @ (vmin - p))), vmin


def closest_boxlist(p, boxes, W):           # boxesND.jl:72-81; boxes = [(lo, hi), ...]
    d2min, vmin, kmin = _m.inf, _np.asarray(p, dtype=float), -1
    for k, (lo, hi) in enumerate(boxes):
        out = closest_box(p, lo, hi, W)
        if out is None:
            continue                        # counted as a failure by the callers of this file
        d2, v = out
        if d2 < d2min:
            d2min, vmin, kmin = d2, v, k
    return d2min, vmin, kmin


def closeR_boxlist(p, boxes, W, r2):        # boxesND.jl:83-86 (sort! by=first is stable)
    cps = [(closest_box(p, lo, hi, W), k) for k, (lo, hi) in enumerate(boxes)]
    return sorted([c + (k,) for c, k in cps if c is not None and c[0] < r2], key=lambda c: c[0])


def closest_circle(p, c, r, W=None):        # SAT2D.jl:208-238
    p = _np.asarray(p, dtype=float); c = _np.asarray(c, dtype=float)
    if W is None:
        xmin = c + r * (p - c) / _np.linalg.norm(p - c)
        return float((p - xmin) @ (p - xmin)), xmin
    vals, vecs = _np.linalg.eigh(_np.asarray(W, dtype=float))
    ctop = p - c
    v1, v2 = vecs[:, 0], vecs[:, 1]
    p1, p2 = v1 @ ctop, v2 @ ctop
    s1, s2 = vals
    lam = 1.0
    f = (p1 * s1 / (lam + s1)) ** 2 + (p2 * s2 / (lam + s2)) ** 2 - r ** 2
    it = 0
    while abs(f) > 1e-8:
        it += 1
        if it > 200 or f != f:              # the reference's loops are unbounded; the restatements stop here and report it
            return None
        fp = -2 / (lam + s1) * (p1 * s1 / (lam + s1)) ** 2 + -2 / (lam + s2) * (p2 * s2 / (lam + s2)) ** 2
        alpha = 1.0
        h = 0
        while True:
            lamnew = lam - alpha * f / fp
            fnew = (p1 * s1 / (lamnew + s1)) ** 2 + (p2 * s2 / (lamnew + s2)) ** 2 - r ** 2
            if abs(fnew) < abs(f):
                break
            alpha /= 2
            h += 1
            if h > 64:
                return None
        f = fnew
        lam = lamnew
    xmin = c + v1 * p1 * s1 / (lam + s1) + v2 * p2 * s2 / (lam + s2)
    return float(s1 * (p1 - p1 * s1 / (lam + s1)) ** 2 + s2 * (p2 - p2 * s2 / (lam + s2)) ** 2), xmin


def closest_polypts(p, points):             # SAT2D.jl:240-254
    p = _np.asarray(p, dtype=float); points = [_np.asarray(q, dtype=float) for q in points]
    N = len(points)
    d2min, vmin = _m.inf, points[0]
    for i in range(N):
        nxt = points[(i + 1) % N]
        edge = nxt - points[i]
        x = (edge @ (p - points[i])) / (edge @ edge)
        v = points[i] if x < 0 else (points[i] + x * edge if x < 1 else nxt)
        d2 = float((p - v) @ (p - v))
        if d2 < d2min:
            d2min, vmin = d2, v
    return d2min, vmin


def closest_polygon(p, points, W=None):     # SAT2D.jl:239, 255-259
    if W is None:
        return closest_polypts(p, points)
    p = _np.asarray(p, dtype=float)
    L = _np.linalg.cholesky(W).T
    xmin = _np.linalg.inv(L) @ closest_polypts(L @ p, [L @ _np.asarray(q, dtype=float) for q in points])[1]
    return float((xmin - p) @ (W @ (xmin - p))), xmin


def closest_compound(p, shapes, W=None):    # SAT2D.jl:260-279; shapes as in oracle.Shapes2D
    d2min, vmin, kmin = _m.inf, _np.zeros(2), -1
    for k, s in enumerate(shapes):
        out = closest_circle(p, s[1], s[2], W) if s[0] == "circle" else closest_polygon(p, s[1], W)
        if out is None:
            continue
        d2, v = out
        if d2 < d2min:
            d2min, vmin, kmin = d2, v, k
    return d2min, vmin, kmin


def closeR_compound(p, shapes, W, r2):      # SAT2D.jl:281-285
    out = []
    for k, s in enumerate(shapes):
        cp = closest_circle(p, s[1], s[2], W) if s[0] == "circle" else closest_polygon(p, s[1], W)
        if cp is not None and cp[0] < r2:
            out.append((cp[0], cp[1], k))
    return sorted(out, key=lambda c: c[0])
