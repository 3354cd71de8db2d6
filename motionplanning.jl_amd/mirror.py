"""Host-side mirror of the reference's Julia plugin surface for the hot path (same names, argument meaning and
error behaviour), written in Python because no Julia toolchain exists in the build image.  Everything that
computes goes through libmpfmt.so (HIP); there is no CPU fallback.

Julia's mutating-function bang becomes a trailing underscore: fmtstar! -> fmtstar_, sample_free! -> sample_free_,
inball! -> inball_.  Indices are 1-based like the reference's (tree `A`, `path`, SparseVector indices).

Reference files mirrored: src/statespaces.jl (BoundedStateSpace, volume, dim, sample_space, is_free_state,
is_free_motion), src/statespaces/geometric.jl (UnitHypercube, BoundedEuclideanStateSpace),
src/statespaces/linearquadratic.jl (DoubleIntegrator), src/statespaces/simplecars.jl (DubinsQuasiMetricSpace, ReedsSheppMetricSpace),
src/collisioncheckers/boxesND.jl (BoxBounds,
PointRobotNDBoxes, inflate, addobstacle, addblocker), src/goals.jl (RectangleGoal, BallGoal, PointGoal, StateGoal),
src/nearneighbors.jl (MetricNN, QuasiMetricNN, inball, inball!, ImmutableNNC, addpoints), src/problems.jl
(MPProblem, MPSolution, clearsamples!), src/sampling.jl (sample_free!), src/planners/fmt.jl (fmtstar!).
"""
import math
import time
import warnings

import numpy as np

from . import _lib
from ._lib import Context, MPFMTError


# ---- metrics / state spaces (src/statespaces.jl:29-42, geometric.jl:10-12, linearquadratic.jl:28-53) ----------
class Euclidean:
    pass


class LinearQuadratic:
    """Double-integrator LQ quasi-metric: R = rho*I, cost radius cmax (linearquadratic.jl:28-39)."""

    def __init__(self, m, rho=1.0, cmax=1.0):
        self.m, self.rho, self.cmax = int(m), float(rho), float(cmax)


class BoundedStateSpace:
    def __init__(self, lo, hi, dist, workspace_dim=None):
        self.lo = np.asarray(lo, dtype=np.float64)
        self.hi = np.asarray(hi, dtype=np.float64)
        self.dist = dist
        self.workspace_dim = len(self.lo) if workspace_dim is None else int(workspace_dim)


def BoundedEuclideanStateSpace(lo, hi):
    return BoundedStateSpace(lo, hi, Euclidean())


def UnitHypercube(d):
    return BoundedEuclideanStateSpace(np.zeros(d), np.ones(d))


def DoubleIntegrator(d, lo=None, hi=None, vmax=1.5, r=1.0):
    lo = np.zeros(d) if lo is None else np.asarray(lo, dtype=np.float64)
    hi = np.ones(d) if hi is None else np.asarray(hi, dtype=np.float64)
    return BoundedStateSpace(np.concatenate([lo, -vmax * np.ones(d)]), np.concatenate([hi, vmax * np.ones(d)]),
                             LinearQuadratic(d, rho=r), workspace_dim=d)


class DubinsExact:
    """Exact Dubins length for turning radius r and speed s, chopped by the Euclidean lower bound on (x, y)
    (simplecars.jl:15-21,32-38)."""

    def __init__(self, r=1.0, s=1.0):
        self.r, self.s = float(r), float(s)


def DubinsQuasiMetricSpace(r, s=1.0, lo=(0.0, 0.0), hi=(1.0, 1.0)):
    """SE2 states (x, y, theta) with theta in [0, 2pi]; workspace = (x, y)   (simplecars.jl:32-38)."""
    return BoundedStateSpace(np.array([lo[0], lo[1], 0.0]), np.array([hi[0], hi[1], 2 * math.pi]), DubinsExact(r, s), workspace_dim=2)


class ReedsSheppExact:
    """Exact Reeds-Shepp length (the car may reverse) for turning radius r and speed s   (simplecars.jl:5-12,23)."""

    def __init__(self, r=1.0, s=1.0):
        self.r, self.s = float(r), float(s)


def ReedsSheppMetricSpace(r, s=1.0, lo=(0.0, 0.0), hi=(1.0, 1.0)):
    """SE2 states with the chopped Reeds-Shepp metric; workspace = (x, y)   (simplecars.jl:29-34)."""
    return BoundedStateSpace(np.array([lo[0], lo[1], 0.0]), np.array([hi[0], hi[1], 2 * math.pi]), ReedsSheppExact(r, s), workspace_dim=2)


def SS_WS(SS):
    return SS.workspace_dim


def volume(SS):
    return float(np.prod(SS.hi - SS.lo))


def dim(SS):
    return len(SS.lo)


def sample_space(SS, rng, n=1):
    """lo + rand .* (hi - lo)   (statespaces.jl:40), n states at once."""
    return SS.lo + rng.random((n, len(SS.lo))) * (SS.hi - SS.lo)


def setup_steering(SS, r):
    if isinstance(SS.dist, LinearQuadratic):
        SS.dist.cmax = float(r)


# ---- collision checker (src/collisioncheckers/boxesND.jl) ----------------------------------------------------------
class BoxBounds:
    def __init__(self, lo, hi=None):
        if hi is None:                       # BoxBounds(lohi::Matrix) = (lohi[:,1], lohi[:,2])
            lohi = np.asarray(lo, dtype=np.float64)
            lo, hi = lohi[:, 0], lohi[:, 1]
        self.lo = np.asarray(lo, dtype=np.float64)
        self.hi = np.asarray(hi, dtype=np.float64)


class PointRobotNDBoxes:
    """Point robot among N-d boxes; `count` = number of segment checks asked for (boxesND.jl:15-28)."""

    def __init__(self, boxes):
        self.boxes = [b if isinstance(b, BoxBounds) else BoxBounds(b) for b in boxes]
        self.count = 0
        self._ctx = None
        self._ss = None

    def lohi(self):
        if not self.boxes:
            return np.zeros((0, 2, 0))
        return np.stack([np.stack([b.lo, b.hi]) for b in self.boxes])

    def _bind(self, ctx, SS):
        dw = SS.workspace_dim
        lohi = self.lohi() if self.boxes else np.zeros((0, 2, dw))
        ctx.upload_boxes(lohi, SS.lo, SS.hi, dw=dw)
        self._ctx, self._ss = ctx, SS

    def _bind_workspace(self, ctx, SS):
        ctx.upload_boxes(self.lohi() if self.boxes else np.zeros((0, 2, SS.workspace_dim)), None, None, dw=SS.workspace_dim)
        self._ctx = None

    def inflate(self, eps):
        return PointRobotNDBoxes([BoxBounds(b.lo - eps, b.hi + eps) for b in self.boxes]) if eps > 0 else self

    def addobstacle(self, o):
        return PointRobotNDBoxes(self.boxes + [o if isinstance(o, BoxBounds) else BoxBounds(o)])

    def addblocker(self, v, r):
        v = np.asarray(v, dtype=np.float64)
        return self.addobstacle(BoxBounds(v - r, v + r))


# ---- 2-D shapes and the SAT point robot (src/collisioncheckers/SAT2D.jl, robots2D.jl) ---------------------------------------
class Circle:
    def __init__(self, c, r):
        if r <= 0:
            raise ValueError("Radius must be positive")                      # SAT2D.jl:22
        self.c, self.r = (float(c[0]), float(c[1])), float(r)

    def parts(self):
        return [("circle", self.c, self.r)]


class Polygon:
    def __init__(self, points):
        if len(points) < 3:
            raise ValueError("Polygons need at least 3 points! Try Line?")    # SAT2D.jl:42
        self.points = [(float(p[0]), float(p[1])) for p in points]

    def parts(self):
        return [("polygon", self.points)]


def Box2D(xr, yr):                                                            # SAT2D.jl:59-62
    return Polygon([(xr[0], yr[0]), (xr[1], yr[0]), (xr[1], yr[1]), (xr[0], yr[1])])


class Compound2D:
    """Compound2D(parts...) -- nested compounds are flattened (every basic test begins with its own AABB check)."""

    def __init__(self, *parts):
        if len(parts) == 1 and isinstance(parts[0], (list, tuple)):
            parts = tuple(parts[0])
        self._parts = list(parts)

    def parts(self):
        return [q for P in self._parts for q in P.parts()]


class PointRobot2D:
    """Point robot among 2-D shapes; `count` = segment checks asked for (robots2D.jl:5-14)."""

    def __init__(self, obstacles):
        self.obstacles = obstacles if isinstance(obstacles, Compound2D) else Compound2D(obstacles)
        self.count = 0
        self._ctx = None
        self._ss = None

    def _bind(self, ctx, SS):
        if SS.workspace_dim != 2:
            raise ValueError("PointRobot2D needs a 2-D workspace")
        ctx.upload_shapes2d(self.obstacles.parts(), SS.lo[:2], SS.hi[:2])
        if dim(SS) != 2:                                     # a steering space over the 2-D world (double integrator, SE2 cars)
            ctx.set_state_bounds(SS.lo, SS.hi)
        self._ctx, self._ss = ctx, SS

    def _bind_workspace(self, ctx, SS):
        ctx.upload_shapes2d(self.obstacles.parts(), None, None)
        self._ctx = None

    def addobstacle(self, o):                                                 # robots2D.jl:23
        return PointRobot2D(Compound2D(self.obstacles, o))

    def addblocker(self, p, r):                                               # robots2D.jl:24
        return self.addobstacle(Circle(p, r))


def is_free_state(v, CC, SS, ctx):
    """in_state_space(v, SS) && is_free_state(state2workspace(v), CC)   (statespaces.jl:151-152); v: (d,) or (n, d)."""
    V = np.atleast_2d(np.asarray(v, dtype=np.float64))
    if CC._ctx is not ctx or CC._ss is not SS:
        CC._bind(ctx, SS)
    if SS.workspace_dim == V.shape[1]:
        out = _lib.unpack_bits(ctx.states_free(V), len(V))
    else:                                                   # workspace = leading coordinates (OutputMatrix [I 0])
        inb = np.all((SS.lo <= V) & (V <= SS.hi), axis=1)
        CC._bind_workspace(ctx, SS)                           # the checker alone, no state-space bounds (they were applied above)
        out = _lib.unpack_bits(ctx.states_free(np.ascontiguousarray(V[:, :SS.workspace_dim])), len(V)) & inb
        CC._bind(ctx, SS)
    return bool(out[0]) if np.ndim(v) == 1 else out


def is_free_motion(v, w, CC, SS, ctx):
    """Euclidean space: in_state_space(v) && segment test (statespaces.jl:153-158, geometric.jl:20); counts like
    boxesND.jl:26.  v, w: (d,) or (n, d)."""
    V = np.atleast_2d(np.asarray(v, dtype=np.float64))
    W = np.atleast_2d(np.asarray(w, dtype=np.float64))
    if not isinstance(SS.dist, Euclidean):
        raise NotImplementedError("scalar is_free_motion is provided for the Euclidean space; use fmtstar_ for LQ")
    if CC._ctx is not ctx or CC._ss is not SS:
        CC._bind(ctx, SS)
    CC.count += int(np.sum(np.all((SS.lo <= V) & (V <= SS.hi), axis=1)))
    out = _lib.unpack_bits(ctx.motions_free(V, W), len(V))
    return bool(out[0]) if np.ndim(v) == 1 else out


# ---- closest obstacle points (boxesND.jl:33-34,61-86; robots2D.jl:25-26; SAT2D.jl:208-285) ------------------------------
def _closest_ctx(CC, ctx, d):
    ctx = CC._ctx if ctx is None else ctx
    if ctx is None:
        raise ValueError("the collision checker is not bound to a device context yet: pass ctx=")
    if CC._ctx is not ctx:
        if isinstance(CC, PointRobot2D):
            ctx.upload_shapes2d(CC.obstacles.parts(), None, None)
        else:
            ctx.upload_boxes(CC.lohi(), None, None, dw=d)
        CC._ctx, CC._ss = ctx, None
    return ctx


def closest(p, CC, W=None, ctx=None):
    """closest(p, CC, W): (d2min, vmin) for p (d,), arrays for p (n, d).  Raises where the reference throws (bvls returned
    `nothing` for some box)."""
    Pm = np.atleast_2d(np.asarray(p, dtype=np.float64))
    ctx = _closest_ctx(CC, ctx, Pm.shape[1])
    d2, v, _, fails = ctx.closest(Pm, W)
    if fails:
        raise RuntimeError("bvls exhausted its iterations on %d (point, box) pair(s): the reference throws here (bvls.jl:67)" % fails)
    return (float(d2[0]), v[0]) if np.ndim(p) == 1 else (d2, v)


def closeR(p, CC, W, r2, ctx=None):
    """closeR(p, CC, W, r2): [(d2, v), ...] ascending for p (d,); a list of such lists for p (n, d)."""
    Pm = np.atleast_2d(np.asarray(p, dtype=np.float64))
    ctx = _closest_ctx(CC, ctx, Pm.shape[1])
    ptr, _, d2, v, fails = ctx.closeR(Pm, W, r2)
    if fails:
        raise RuntimeError("bvls exhausted its iterations on %d (point, box) pair(s): the reference throws here (bvls.jl:67)" % fails)
    out = [[(float(d2[e]), v[e]) for e in range(ptr[i] - 1, ptr[i + 1] - 1)] for i in range(len(Pm))]
    return out[0] if np.ndim(p) == 1 else out


# ---- goals (src/goals.jl) -----------------------------------------------------------------------------------------------
class RectangleGoal:
    kind = _lib.GOAL_RECT

    def __init__(self, lo, hi):
        self.lo, self.hi = np.asarray(lo, dtype=np.float64), np.asarray(hi, dtype=np.float64)

    def params(self):
        return np.concatenate([self.lo, self.hi])

    def sample(self, rng):
        return self.lo + (self.hi - self.lo) * rng.random(len(self.lo))


class BallGoal:
    kind = _lib.GOAL_BALL

    def __init__(self, center, radius):
        self.center, self.radius = np.asarray(center, dtype=np.float64), float(radius)

    def params(self):
        return np.concatenate([self.center, [self.radius]])

    def sample(self, rng):
        while True:                                          # goals.jl:101-108
            v = self.center + 2 * self.radius * (rng.random(len(self.center)) - 0.5)
            if np.linalg.norm(v - self.center) <= self.radius:
                return v


class PointGoal:
    """ConvexHullWorkspaceGoal with one point (goals.jl:45): is_goal_pt = exact equality."""
    kind = _lib.GOAL_POINT

    def __init__(self, pt):
        self.pt = np.asarray(pt, dtype=np.float64)

    def params(self):
        return self.pt

    def sample(self, rng):
        return self.pt.copy()


StateGoal = PointGoal                                        # ConvexHullStateSpaceGoal([s]) (goals.jl:68)


def sample_goal(G, SS, rng):
    """Workspace goals are lifted to a state by completing the other coordinates with a space sample
    (workspace2state, statespaces.jl:62-70)."""
    g = G.sample(rng)
    if len(g) == len(SS.lo):
        return g
    v = sample_space(SS, rng, 1)[0]
    v[:len(g)] = g
    return v


# ---- sample sets (src/nearneighbors.jl) -----------------------------------------------------------------------------------
class ImmutableNNC:
    """SparseMatrixCSC neighbour cache (nearneighbors.jl:23-27): colptr/rowval 1-based, nzval."""

    def __init__(self, colptr, rowval, nzval, r):
        self.colptr, self.rowval, self.nzval, self.r = colptr, rowval, nzval, r

    def viewcol(self, v):
        a, b = self.colptr[v - 1] - 1, self.colptr[v] - 1
        return self.rowval[a:b], self.nzval[a:b]


class MetricNN:
    """SampleSet for symmetric distances (nearneighbors.jl:62-74); DS is the device context holding the samples."""

    def __init__(self, V, dist, init, ctx=None):
        self.V = np.ascontiguousarray(np.atleast_2d(V), dtype=np.float64)
        self.dist, self.init = dist, np.asarray(init, dtype=np.float64)
        self.cache = None
        self.DS = ctx if ctx is not None else Context(0)     # helper_data_structures (geometric.jl:14)
        self.DS.upload_samples(self.V)

    def __len__(self):
        return len(self.V)

    def __getitem__(self, i):
        return self.V[i - 1] if i > 0 else self.init         # nearneighbors.jl:111


QuasiMetricNN = MetricNN


def addpoints(NN, W):
    return MetricNN(np.concatenate([NN.V, np.atleast_2d(W)]), NN.dist, NN.init, NN.DS)


def inball(NN, v, r):
    """inball(V, dist, DS, v, r) -> (inds, ds): ascending 1-based indices, self excluded (nearneighbors.jl:179-183)."""
    return NN.DS.rdisc_query(v, r)


def inball_(NN, v, r, f=None):
    """inball!(NN, v, r[, f]): cached query; with an ImmutableNNC installed it is `viewcol` (nearneighbors.jl:120-136)."""
    if NN.cache is not None and NN.cache.r == r:
        inds, ds = NN.cache.viewcol(v)
    else:
        inds, ds = inball(NN, v, r)
    if f is not None:                                         # filter_neighborhood (nearneighbors.jl:104-107)
        keep = np.asarray(f, dtype=bool)[inds - 1]
        inds, ds = inds[keep], ds[keep]
    return inds, ds


def build_cache_(NN, r):
    """Whole r-disc graph at once -> ImmutableNNC (what the eager GPU path installs)."""
    colptr, rowval, nzval = NN.DS.rdisc_graph(r)
    NN.cache = ImmutableNNC(colptr, rowval, nzval, r)
    return NN.cache


# ---- problem (src/problems.jl) -----------------------------------------------------------------------------------------------
class MPSolution:
    def __init__(self, status, cost, elapsed, metadata):
        self.status, self.cost, self.elapsed, self.metadata = status, cost, elapsed, metadata


class MPProblem:
    def __init__(self, SS, init, goal, CC, ctx=None):
        self.SS, self.goal, self.CC = SS, goal, CC
        self.init = np.asarray(init, dtype=np.float64)
        self.ctx = ctx if ctx is not None else Context(0)
        self.V = MetricNN(self.init[None, :], SS.dist, self.init, self.ctx)      # defaultNN (statespaces.jl:163-170)
        self.status = "not yet solved"
        self.solution = None


def clearsamples_(P):
    P.V = MetricNN(P.init[None, :], P.SS.dist, P.init, P.ctx)


# ---- sampling (src/sampling.jl) ------------------------------------------------------------------------------------------------
def sample_free_goal(P, rng):
    while True:
        v = sample_goal(P.goal, P.SS, rng)
        if is_free_state(v, P.CC, P.SS, P.ctx):
            return v


def sample_free_(P, N, ensure_goal=True, ensure_goal_ct=5, rng=None, seed=None):
    """sample_free!(P, N): N new free samples (rejection sampling, batched: candidates are validity-checked on the
    GPU in blocks), goal samples written into the tail (sampling.jl:11-45).  Returns volume(SS).
    With `seed` (and an empty sample set, a Euclidean space, a Rectangle/Ball/Point goal) the whole loop runs on the
    device from a counter-based stream (`Context.sample_free`, mpfmt_sample_free); otherwise candidates come from `rng`."""
    rng = np.random.default_rng() if rng is None else rng
    if N <= 0:
        return volume(P.SS)
    has_init = len(P.V.V) > 0 and np.array_equal(P.V.V[0], P.init)                 # sampling.jl:15-20
    if seed is not None and len(P.V.V) <= 1 and isinstance(P.SS.dist, Euclidean) and hasattr(P.goal, "kind"):
        P.CC._bind(P.ctx, P.SS)
        W, _ = P.ctx.sample_free(seed, N, init=None if has_init else P.init, goal_kind=P.goal.kind,
                                 goal_params=P.goal.params(), goal_ct=ensure_goal_ct if ensure_goal else 0)
        P.V = addpoints(P.V, W)
        return volume(P.SS)
    W = np.empty((N, dim(P.SS)))
    have = 0
    if not (len(P.V.V) > 0 and np.array_equal(P.V.V[0], P.init)):
        W[0] = P.init
        have = 1
    while have < N:
        cand = sample_space(P.SS, rng, max(1024, 2 * (N - have)))
        ok = is_free_state(cand, P.CC, P.SS, P.ctx)
        good = cand[ok][:N - have]
        W[have:have + len(good)] = good
        have += len(good)
    if ensure_goal:
        for i in range(1, min(ensure_goal_ct, N - 1) + 1):
            W[N - i] = sample_free_goal(P, rng)
    P.V = addpoints(P.V, W)
    return volume(P.SS)


# ---- planner (src/planners/fmt.jl) -------------------------------------------------------------------------------------------------
def fmtstar_(P, N=None, rm=1.0, connections="R", r=0.0, ensure_goal_ct=1, init_idx=1, checkpts=True, rng=None, seed=None,
             band=None):
    """fmtstar!(P, N; rm, connections, r, ensure_goal_ct, init_idx, checkpts)  (fmt.jl:3-119).  Returns
    (status, cost, elapsed) and fills P.solution like the reference.  band = None: the reference's sequential recursion (on the
    host, over GPU-built arrays); band >= 0: the recursion on the device with cost-band batches of that width in units of r
    (mpfmt_*_fmtstar_wavefront; band = 0 expands only exact cost ties together)."""
    t0 = time.time()
    N = len(P.V) if N is None else int(N)
    P.CC.count = 0
    if connections != "R":
        raise ValueError("Connection type must be radial (:R); the k-nearest branch of the reference calls "
                         "undefined functions (fmt.jl:17-19)")
    if r > 0:
        setup_steering(P.SS, r)
    if not is_free_state(P.init, P.CC, P.SS, P.ctx):
        warnings.warn("Initial state is infeasible!")
        P.status = "failed"
        P.solution = MPSolution(P.status, math.inf, time.time() - t0, {})
        return math.inf
    free_volume_ub = sample_free_(P, N - len(P.V), ensure_goal_ct=ensure_goal_ct, rng=rng, seed=seed)
    if r == 0:
        d = dim(P.SS)
        r = rm * 2 * (1 / d * free_volume_ub / (math.pi ** (d / 2) / math.gamma(d / 2 + 1)) * math.log(N) / N) ** (1 / d)
        setup_steering(P.SS, r)
    ctx = P.ctx
    P.CC._bind(ctx, P.SS)
    gkind, gpar = P.goal.kind, P.goal.params()
    if isinstance(P.SS.dist, (DubinsExact, ReedsSheppExact)) and gkind == _lib.GOAL_POINT and len(gpar) == SS_WS(P.SS):
        # PointGoal(pt) is a WORKSPACE goal (goals.jl:45,111-114: state2workspace(v) == pt); for SE2 states the library's POINT kind
        # means an exact state, so the workspace point goes down as the ball of radius 0 around it (norm(v_ws - pt) <= 0)
        gkind, gpar = _lib.GOAL_BALL, np.concatenate([gpar, [0.0]])
    if band is not None:
        bw = float(band) * r
        if isinstance(P.SS.dist, LinearQuadratic):
            res = ctx.di_fmtstar_wavefront(P.SS.dist.rho, r, gkind, gpar, band=bw, init_idx=init_idx, checkpts=checkpts)
        elif isinstance(P.SS.dist, (DubinsExact, ReedsSheppExact)):
            car = "dubins" if isinstance(P.SS.dist, DubinsExact) else "reedsshepp"
            res = ctx.car_fmtstar_wavefront(car, P.SS.dist.r, P.SS.dist.s, r, gkind, gpar, band=bw, init_idx=init_idx,
                                            checkpts=checkpts)
        else:
            res = ctx.fmtstar_wavefront(r, gkind, gpar, band=bw, init_idx=init_idx, checkpts=checkpts)
    elif isinstance(P.SS.dist, LinearQuadratic):
        res = ctx.di_fmtstar(P.SS.dist.rho, r, gkind, gpar, init_idx=init_idx, checkpts=checkpts)
    elif isinstance(P.SS.dist, DubinsExact):
        res = ctx.dubins_fmtstar(P.SS.dist.r, P.SS.dist.s, r, gkind, gpar, init_idx=init_idx, checkpts=checkpts)
    elif isinstance(P.SS.dist, ReedsSheppExact):
        res = ctx.reedsshepp_fmtstar(P.SS.dist.r, P.SS.dist.s, r, gkind, gpar, init_idx=init_idx, checkpts=checkpts)
    else:
        res = ctx.fmtstar(r, gkind, gpar, init_idx=init_idx, checkpts=checkpts)
    P.CC.count = res["collision_checks"]
    P.status = "solved" if res["status"] == 1 else "failed"
    path = res["path"]
    meta = {"radius_multiplier": rm, "collision_checks": res["collision_checks"], "num_samples": N, "cost": res["cost"],
            "cumcost": res["C"][path - 1], "planner": "FMTstar", "solved": res["status"] == 1, "tree": res["A"],
            "path": path, "r": r, "ms_graph": res["ms_graph"], "ms_sweep": res["ms_sweep"], "ms_host_loop": res["ms_host_loop"]}
    P.solution = MPSolution(P.status, res["cost"], time.time() - t0, meta)
    return P.status, P.solution.cost, P.solution.elapsed
