"""ctypes binding of the CPU oracle (oracle/mpfmt_oracle.c).

TEST INFRASTRUCTURE ONLY: may be imported by tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py -- never by the product package.  PARITY UNPINNED (see the
header of mpfmt_oracle.c): the reference has no tests and cannot be run here.

Array conventions (the memory layouts of the reference's Julia objects):
  X     : float64 (N, d) C-contiguous  == Julia d x N column-major == Vector{SVector{d,Float64}}
  lohi  : float64 (M, 2, d)            == Vector{BoxBounds{d}} = [lo(d); hi(d)] per box
  masks : uint64 words, bit e of word e>>6 (LSB first) == BitVector.chunks
Indices are 0-based here (the C ABI of the product is 1-based like Julia; tests convert).
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

c_d_p = C.POINTER(C.c_double)
c_i64_p = C.POINTER(C.c_int64)
c_u64_p = C.POINTER(C.c_uint64)
c_u8_p = C.POINTER(C.c_uint8)


class FmtResult(C.Structure):
    _fields_ = [("status", C.c_int32), ("cost", C.c_double), ("z", C.c_int64),
                ("collision_checks", C.c_int64), ("path_len", C.c_int64), ("nn_queries", C.c_int64)]


def build(force=False, devmath=False):
    so = os.path.join(_HERE, "liboracle_devmath.so" if devmath else "liboracle.so")
    src = os.path.join(_HERE, "mpfmt_oracle.c")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return so


def _declare(L):
    L.orc_sqdist.restype = C.c_double
    L.orc_dist.restype = C.c_double
    L.orc_inball.restype = C.c_int64
    L.orc_rdisc_count.restype = C.c_int64
    L.orc_rdisc_fill.restype = None
    L.orc_kdtree_build.restype = C.c_void_p
    L.orc_kdtree_free.restype = None
    L.orc_kdtree_inball.restype = C.c_int64
    L.orc_expand.restype = C.c_int64
    for f in (L.orc_mp_sin, L.orc_mp_cos, L.orc_mp_atan2, L.orc_mp_acos):
        f.restype = C.c_double
    L.orc_splitmix64.restype = C.c_uint64
    L.orc_stream_uniform.restype = None
    L.orc_euclid_steer.restype = None
    L.orc_euclid_propagate.restype = None
    L.orc_di_pairwise.restype = C.c_int64
    L.orc_fmt_radius.restype = C.c_double
    for f in ("orc_di_cost", "orc_di_dcost", "orc_di_ddcost"):
        getattr(L, f).restype = C.c_double
    return L


_LIB_DEV = None          # the second build (liboracle_devmath.so): the car models' sin / cos / atan2 / acos are the DEVICE's own
_USE_DEV = False


def lib():
    """The oracle library every wrapper below calls: liboracle.so (C library sin / cos / atan2 / acos -- independent of the product), or,
    inside `with device_math():`, liboracle_devmath.so (the product's generated mp_math.h: self-consistency checks only)."""
    global _LIB, _LIB_DEV
    if _USE_DEV:
        if _LIB_DEV is None:
            _LIB_DEV = _declare(C.CDLL(build(devmath=True)))
        return _LIB_DEV
    if _LIB is None:
        _LIB = _declare(C.CDLL(build()))
    return _LIB


class device_math:
    """`with orc.device_math(): ...` -- the car-model functions of the oracle run with the transcendental functions the DEVICE compiles
    (one implementation on both sides: ties between words break identically, so trees and controls can be compared bit for bit).
    A labelled self-consistency check, not a parity claim: outside this block the oracle calls the C library."""
    def __enter__(self):
        global _USE_DEV
        self._was = _USE_DEV
        _USE_DEV = True
        return self

    def __exit__(self, *a):
        global _USE_DEV
        _USE_DEV = self._was


def _d(a):
    return None if a is None else a.ctypes.data_as(c_d_p)


def _i(a):
    return None if a is None else a.ctypes.data_as(c_i64_p)


def _u(a):
    return None if a is None else a.ctypes.data_as(c_u64_p)


def _X(X):
    X = np.ascontiguousarray(X, dtype=np.float64)
    assert X.ndim == 2
    return X, X.shape[0], X.shape[1]


def _boxes(lohi, d):
    if lohi is None:
        return np.zeros((0, 2, d)), 0
    lohi = np.ascontiguousarray(lohi, dtype=np.float64)
    if lohi.size == 0:
        return np.zeros((0, 2, d)), 0
    assert lohi.ndim == 3 and lohi.shape[1] == 2
    return lohi, lohi.shape[0]


def _vec(a):
    return None if a is None else np.ascontiguousarray(a, dtype=np.float64)


def nwords(n):
    return (n + 63) // 64


def unpack(mask, n):
    """uint64 LSB-first words -> bool array of length n."""
    b = np.unpackbits(np.ascontiguousarray(mask).view(np.uint8), bitorder="little")
    return b[:n].astype(bool)


def pack(bits):
    bits = np.asarray(bits, dtype=bool)
    n = bits.size
    pad = np.zeros(nwords(n) * 64, dtype=np.uint8)
    pad[:n] = bits
    return np.packbits(pad, bitorder="little").view(np.uint64)


def sqdist(a, b):
    a = _vec(a); b = _vec(b)
    return lib().orc_sqdist(_d(a), _d(b), C.c_int32(a.size))


def inball(X, v, r, mode=0):
    X, N, d = _X(X)
    inds = np.empty(N, dtype=np.int64); ds = np.empty(N, dtype=np.float64)
    k = lib().orc_inball(_d(X), C.c_int64(N), C.c_int32(d), C.c_int64(v), C.c_double(r), C.c_int32(mode),
                         _i(inds), _d(ds), C.c_int64(N))
    return inds[:k].copy(), ds[:k].copy()


def rdisc_graph(X, r, mode=0):
    """Full r-disc graph as 0-based CSC (colptr, rowval, nzval)."""
    X, N, d = _X(X)
    colptr = np.empty(N + 1, dtype=np.int64)
    nnz = lib().orc_rdisc_count(_d(X), C.c_int64(N), C.c_int32(d), C.c_double(r), C.c_int32(mode), _i(colptr))
    rowval = np.empty(max(nnz, 1), dtype=np.int64); nzval = np.empty(max(nnz, 1), dtype=np.float64)
    lib().orc_rdisc_fill(_d(X), C.c_int64(N), C.c_int32(d), C.c_double(r), C.c_int32(mode), _i(colptr), _i(rowval), _d(nzval))
    return colptr, rowval[:nnz], nzval[:nnz]


class KDTree:
    def __init__(self, X):
        self.X, self.N, self.d = _X(X)
        self._t = C.c_void_p(lib().orc_kdtree_build(_d(self.X), C.c_int64(self.N), C.c_int32(self.d)))
        self._inds = np.empty(max(self.N, 1), dtype=np.int64)
        self._ds = np.empty(max(self.N, 1), dtype=np.float64)

    def inball(self, v, r):
        k = lib().orc_kdtree_inball(self._t, C.c_int64(v), C.c_double(r), _i(self._inds), _d(self._ds), C.c_int64(self.N))
        return self._inds[:k].copy(), self._ds[:k].copy()

    def inball_count_only(self, v, r):
        return lib().orc_kdtree_inball(self._t, C.c_int64(v), C.c_double(r), _i(self._inds), _d(self._ds), C.c_int64(self.N))

    def __del__(self):
        try:
            lib().orc_kdtree_free(self._t)
        except Exception:
            pass


def point_free_boxes(v, lohi):
    v = _vec(v); lohi, M = _boxes(lohi, v.size)
    return bool(lib().orc_point_free_boxes(_d(v), _d(lohi), C.c_int32(M), C.c_int32(v.size)))


def motion_free_boxes(v, w, lohi):
    v = _vec(v); w = _vec(w); lohi, M = _boxes(lohi, v.size)
    return bool(lib().orc_motion_free_boxes(_d(v), _d(w), _d(lohi), C.c_int32(M), C.c_int32(v.size)))


def box_phases(v, w, lo, hi):
    """(broadphase_free, narrow_free) for one box (boxesND.jl:44-51)."""
    v = _vec(v); w = _vec(w); lo = _vec(lo); hi = _vec(hi)
    d = C.c_int32(v.size)
    return (bool(lib().orc_box_broadphase_free(_d(v), _d(w), _d(lo), _d(hi), d)),
            bool(lib().orc_box_narrow_free(_d(v), _d(w), _d(lo), _d(hi), d)))


def is_free_state(v, lohi, ss_lo=None, ss_hi=None):
    v = _vec(v); lohi, M = _boxes(lohi, v.size); ss_lo = _vec(ss_lo); ss_hi = _vec(ss_hi)
    return bool(lib().orc_is_free_state(_d(v), C.c_int32(v.size), _d(lohi), C.c_int32(M), _d(ss_lo), _d(ss_hi)))


def is_free_motion(v, w, lohi, ss_lo=None, ss_hi=None):
    v = _vec(v); w = _vec(w); lohi, M = _boxes(lohi, v.size); ss_lo = _vec(ss_lo); ss_hi = _vec(ss_hi)
    return bool(lib().orc_is_free_motion(_d(v), _d(w), C.c_int32(v.size), _d(lohi), C.c_int32(M), _d(ss_lo), _d(ss_hi)))


def points_free(X, lohi, ss_lo=None, ss_hi=None, idx=None):
    X, N, d = _X(X); lohi, M = _boxes(lohi, d); ss_lo = _vec(ss_lo); ss_hi = _vec(ss_hi)
    if idx is not None:
        idx = np.ascontiguousarray(idx, dtype=np.int64)
    n = N if idx is None else idx.size
    mask = np.zeros(max(nwords(n), 1), dtype=np.uint64)
    lib().orc_points_free(_d(X), C.c_int64(N), C.c_int32(d), _i(idx), C.c_int64(n), _d(lohi), C.c_int32(M),
                          _d(ss_lo), _d(ss_hi), _u(mask))
    return mask[:nwords(n)]


def edges_free(X, src, dst, lohi, ss_lo=None, ss_hi=None):
    X, N, d = _X(X); lohi, M = _boxes(lohi, d); ss_lo = _vec(ss_lo); ss_hi = _vec(ss_hi)
    src = np.ascontiguousarray(src, dtype=np.int64); dst = np.ascontiguousarray(dst, dtype=np.int64)
    E = src.size
    mask = np.zeros(max(nwords(E), 1), dtype=np.uint64)
    lib().orc_edges_free(_d(X), C.c_int64(N), C.c_int32(d), _i(src), _i(dst), C.c_int64(E), _d(lohi), C.c_int32(M),
                         _d(ss_lo), _d(ss_hi), _u(mask))
    return mask[:nwords(E)]


def graph_edges_free(X, colptr, rowval, lohi, ss_lo=None, ss_hi=None):
    X, N, d = _X(X); lohi, M = _boxes(lohi, d); ss_lo = _vec(ss_lo); ss_hi = _vec(ss_hi)
    colptr = np.ascontiguousarray(colptr, dtype=np.int64); rowval = np.ascontiguousarray(rowval, dtype=np.int64)
    nnz = int(colptr[N])
    mask = np.zeros(max(nwords(nnz), 1), dtype=np.uint64)
    lib().orc_graph_edges_free(_d(X), C.c_int64(N), C.c_int32(d), _i(colptr), _i(rowval), _d(lohi), C.c_int32(M),
                               _d(ss_lo), _d(ss_hi), _u(mask))
    return mask[:nwords(nnz)]


GOAL_RECT, GOAL_BALL, GOAL_POINT = 0, 1, 2


def is_goal_pt(v, kind, g):
    v = _vec(v); g = _vec(g)
    return bool(lib().orc_is_goal_pt(_d(v), C.c_int32(v.size), C.c_int32(kind), _d(g)))


def expand(X, r, W, H, F, Cc, zs, lohi, ss_lo=None, ss_hi=None):
    X, N, d = _X(X); lohi, M = _boxes(lohi, d); ss_lo = _vec(ss_lo); ss_hi = _vec(ss_hi)
    W = np.ascontiguousarray(W, dtype=np.uint64); H = np.ascontiguousarray(H, dtype=np.uint64)
    F = None if F is None else np.ascontiguousarray(F, dtype=np.uint64)
    Cc = _vec(Cc); zs = np.ascontiguousarray(zs, dtype=np.int64)
    xs = np.empty(max(N, 1), dtype=np.int64); ym = np.empty(max(N, 1), dtype=np.int64)
    cm = np.empty(max(N, 1), dtype=np.float64); fr = np.empty(max(N, 1), dtype=np.uint8)
    nx = lib().orc_expand(_d(X), C.c_int64(N), C.c_int32(d), C.c_double(r), _u(W), _u(H), _u(F), _d(Cc),
                          _i(zs), C.c_int64(zs.size), _d(lohi), C.c_int32(M), _d(ss_lo), _d(ss_hi),
                          _i(xs), _i(ym), _d(cm), fr.ctypes.data_as(c_u8_p))
    return xs[:nx].copy(), ym[:nx].copy(), cm[:nx].copy(), fr[:nx].astype(bool)


def fmtstar(X, r, goal_kind, goal, lohi, ss_lo=None, ss_hi=None, init_idx=0, checkpts=True, nn_mode=0):
    X, N, d = _X(X); lohi, M = _boxes(lohi, d); ss_lo = _vec(ss_lo); ss_hi = _vec(ss_hi); goal = _vec(goal)
    A = np.empty(N, dtype=np.int64); Cc = np.empty(N, dtype=np.float64); path = np.empty(N, dtype=np.int64)
    res = FmtResult()
    rc = lib().orc_fmtstar(_d(X), C.c_int64(N), C.c_int32(d), C.c_double(r), C.c_int64(init_idx), C.c_int32(int(checkpts)),
                           C.c_int32(goal_kind), _d(goal), _d(lohi), C.c_int32(M), _d(ss_lo), _d(ss_hi),
                           C.c_int32(nn_mode), _i(A), _d(Cc), _i(path), C.byref(res))
    return dict(rc=rc, status=int(res.status), cost=float(res.cost), z=int(res.z),
                collision_checks=int(res.collision_checks), nn_queries=int(res.nn_queries),
                A=A, C=Cc, path=path[:res.path_len].copy())


def fmtstar_graph(X, colptr, rowval, nzval, free_mask, Fmask, goal_kind, goal, lohi, ss_lo=None, ss_hi=None, init_idx=0):
    X, N, d = _X(X); lohi, M = _boxes(lohi, d); ss_lo = _vec(ss_lo); ss_hi = _vec(ss_hi); goal = _vec(goal)
    colptr = np.ascontiguousarray(colptr, dtype=np.int64); rowval = np.ascontiguousarray(rowval, dtype=np.int64)
    nzval = _vec(nzval)
    free_mask = None if free_mask is None else np.ascontiguousarray(free_mask, dtype=np.uint64)
    Fmask = None if Fmask is None else np.ascontiguousarray(Fmask, dtype=np.uint64)
    A = np.empty(N, dtype=np.int64); Cc = np.empty(N, dtype=np.float64); path = np.empty(N, dtype=np.int64)
    res = FmtResult()
    rc = lib().orc_fmtstar_graph(_d(X), C.c_int64(N), C.c_int32(d), C.c_int64(init_idx), _i(colptr), _i(rowval), _d(nzval),
                                 _u(free_mask), _u(Fmask), C.c_int32(goal_kind), _d(goal), _d(lohi), C.c_int32(M),
                                 _d(ss_lo), _d(ss_hi), _i(A), _d(Cc), _i(path), C.byref(res))
    return dict(rc=rc, status=int(res.status), cost=float(res.cost), z=int(res.z),
                collision_checks=int(res.collision_checks), A=A, C=Cc, path=path[:res.path_len].copy())


def mp_sin(x):
    return lib().orc_mp_sin(C.c_double(x))


def mp_cos(x):
    return lib().orc_mp_cos(C.c_double(x))


def mp_atan2(y, x):
    return lib().orc_mp_atan2(C.c_double(y), C.c_double(x))


def mp_acos(x):
    return lib().orc_mp_acos(C.c_double(x))


def splitmix64(seed, i):
    return int(lib().orc_splitmix64(C.c_uint64(seed), C.c_uint64(i)))


def stream_uniform(seed, n, offset=0):
    u = np.empty(n)
    lib().orc_stream_uniform(C.c_uint64(seed), C.c_uint64(offset), C.c_int64(n), _d(u))
    return u


def euclid_steer(v, w):
    v = _vec(v); w = _vec(w)
    t = C.c_double(); u = np.empty(v.size)
    lib().orc_euclid_steer(_d(v), _d(w), C.c_int32(v.size), C.byref(t), _d(u))
    return t.value, u


def euclid_propagate(v, t, u, s=None):
    v = _vec(v); u = _vec(u); out = np.empty(v.size)
    lib().orc_euclid_propagate(_d(v), C.c_int32(v.size), C.c_double(t), _d(u), C.c_int32(0 if s is None else 1),
                               C.c_double(0.0 if s is None else s), _d(out))
    return out


def fmt_wavefront_graph(X, colptr, rowval, nzval, free_mask, Fmask, goal_kind, goal, lohi, ss_lo=None, ss_hi=None, init_idx=0,
                        band=0.0, single=False):
    """The batched (wavefront) form of the loop on a prebuilt graph: checker of mpfmt_fmtstar_wavefront."""
    X, N, d = _X(X); lohi, M = _boxes(lohi, d); ss_lo = _vec(ss_lo); ss_hi = _vec(ss_hi); goal = _vec(goal)
    colptr = np.ascontiguousarray(colptr, dtype=np.int64); rowval = np.ascontiguousarray(rowval, dtype=np.int64)
    nzval = _vec(nzval)
    free_mask = None if free_mask is None else np.ascontiguousarray(free_mask, dtype=np.uint64)
    Fmask = None if Fmask is None else np.ascontiguousarray(Fmask, dtype=np.uint64)
    A = np.empty(N, dtype=np.int64); Cc = np.empty(N, dtype=np.float64); path = np.empty(N, dtype=np.int64)
    res = FmtResult(); iters = C.c_int64()
    rc = lib().orc_fmt_wavefront_graph(_d(X), C.c_int64(N), C.c_int32(d), C.c_int64(init_idx), _i(colptr), _i(rowval), _d(nzval),
                                       _u(free_mask), _u(Fmask), C.c_int32(goal_kind), _d(goal), _d(lohi), C.c_int32(M),
                                       _d(ss_lo), _d(ss_hi), C.c_double(band), C.c_int32(int(single)), _i(A), _d(Cc), _i(path),
                                       C.byref(res), C.byref(iters))
    return dict(rc=rc, status=int(res.status), cost=float(res.cost), z=int(res.z), iters=int(iters.value),
                collision_checks=int(res.collision_checks), A=A, C=Cc, path=path[:res.path_len].copy())


def fmt_wavefront_directed(X, gd, colptr, rowval, nzval, free_mask, nseg, Fmask, goal_kind, goal, init_idx=0, init_free=True, band=0.0,
                           single=False):
    """The batched loop over a directed cost graph (CSC = backward sets; forward sets derived here), eager edge bits + nseg."""
    X, N, d = _X(X); goal = _vec(goal)
    colptr = np.ascontiguousarray(colptr, dtype=np.int64); rowval = np.ascontiguousarray(rowval, dtype=np.int64)
    nzval = _vec(nzval)
    nnz = int(colptr[N])
    cols = np.repeat(np.arange(N, dtype=np.int64), np.diff(colptr))
    order = np.lexsort((cols, rowval[:nnz]))                 # rows ascending, targets ascending inside a row
    colidx = np.ascontiguousarray(cols[order])
    rowptr = np.zeros(N + 1, dtype=np.int64)
    np.cumsum(np.bincount(rowval[:nnz], minlength=N), out=rowptr[1:])
    free_mask = np.ascontiguousarray(free_mask, dtype=np.uint64)
    nseg = np.ascontiguousarray(nseg, dtype=np.uint8)
    Fmask = None if Fmask is None else np.ascontiguousarray(Fmask, dtype=np.uint64)
    A = np.empty(N, dtype=np.int64); Cc = np.empty(N, dtype=np.float64); path = np.empty(N, dtype=np.int64)
    res = FmtResult(); iters = C.c_int64()
    rc = lib().orc_fmt_wavefront_directed(_d(X), C.c_int64(N), C.c_int32(d), C.c_int32(gd), C.c_int64(init_idx), C.c_int32(int(init_free)),
                                          _i(colptr), _i(rowval), _d(nzval), _i(rowptr), _i(colidx), _u(free_mask),
                                          nseg.ctypes.data_as(c_u8_p), _u(Fmask), C.c_int32(goal_kind), _d(goal), C.c_double(band),
                                          C.c_int32(int(single)), _i(A), _d(Cc), _i(path), C.byref(res), C.byref(iters))
    return dict(rc=rc, status=int(res.status), cost=float(res.cost), z=int(res.z), iters=int(iters.value),
                collision_checks=int(res.collision_checks), A=A, C=Cc, path=path[:res.path_len].copy())


def fmt_radius(rm, d, vol, N):
    return lib().orc_fmt_radius(C.c_double(rm), C.c_int32(d), C.c_double(vol), C.c_int64(N))


# --- double integrator (LQ) -------------------------------------------------------------------

def di_cost(x0, x1, rho, t):
    x0 = _vec(x0); x1 = _vec(x1)
    return lib().orc_di_cost(_d(x0), _d(x1), C.c_int32(x0.size // 2), C.c_double(rho), C.c_double(t))


def di_dcost(x0, x1, rho, t):
    x0 = _vec(x0); x1 = _vec(x1)
    return lib().orc_di_dcost(_d(x0), _d(x1), C.c_int32(x0.size // 2), C.c_double(rho), C.c_double(t))


def di_ddcost(x0, x1, rho, t):
    x0 = _vec(x0); x1 = _vec(x1)
    return lib().orc_di_ddcost(_d(x0), _d(x1), C.c_int32(x0.size // 2), C.c_double(rho), C.c_double(t))


def di_steer(x0, x1, rho, r):
    x0 = _vec(x0); x1 = _vec(x1)
    cost = C.c_double(); t = C.c_double()
    lib().orc_di_steer(_d(x0), _d(x1), C.c_int32(x0.size // 2), C.c_double(rho), C.c_double(r), C.byref(cost), C.byref(t))
    return cost.value, t.value


def di_state(x0, x1, rho, t, s):
    x0 = _vec(x0); x1 = _vec(x1)
    out = np.empty(x0.size)
    lib().orc_di_state(_d(x0), _d(x1), C.c_int32(x0.size // 2), C.c_double(rho), C.c_double(t), C.c_double(s), _d(out))
    return out


def di_waypoints(x0, x1, rho, r):
    x0 = _vec(x0); x1 = _vec(x1)
    out = np.empty((5, x0.size))
    lib().orc_di_waypoints(_d(x0), _d(x1), C.c_int32(x0.size // 2), C.c_double(rho), C.c_double(r), _d(out))
    return out


def di_is_free_motion(x0, x1, rho, r, lohi, ss_lo=None, ss_hi=None):
    x0 = _vec(x0); x1 = _vec(x1); m = x0.size // 2
    lohi, M = _boxes(lohi, m); ss_lo = _vec(ss_lo); ss_hi = _vec(ss_hi)
    return bool(lib().orc_di_is_free_motion(_d(x0), _d(x1), C.c_int32(m), C.c_double(rho), C.c_double(r),
                                            _d(lohi), C.c_int32(M), _d(ss_lo), _d(ss_hi)))


def di_pairwise(X, rho, r):
    """Sparse DSB cost matrix (column j = sources i that reach j with cost <= r), 0-based CSC + optimal times."""
    X, N, n = _X(X)
    colptr = np.empty(N + 1, dtype=np.int64)
    nnz = lib().orc_di_pairwise(_d(X), C.c_int64(N), C.c_int32(n // 2), C.c_double(rho), C.c_double(r), _i(colptr), None, None, None)
    rowval = np.empty(max(nnz, 1), dtype=np.int64); nzval = np.empty(max(nnz, 1)); tval = np.empty(max(nnz, 1))
    lib().orc_di_pairwise(_d(X), C.c_int64(N), C.c_int32(n // 2), C.c_double(rho), C.c_double(r), _i(colptr), _i(rowval), _d(nzval), _d(tval))
    return colptr, rowval[:nnz], nzval[:nnz], tval[:nnz]


def di_graph_edges_free(X, rho, r, colptr, rowval, lohi, ss_lo=None, ss_hi=None):
    X, N, n = _X(X); m = n // 2
    lohi, M = _boxes(lohi, m); ss_lo = _vec(ss_lo); ss_hi = _vec(ss_hi)
    colptr = np.ascontiguousarray(colptr, dtype=np.int64); rowval = np.ascontiguousarray(rowval, dtype=np.int64)
    nnz = int(colptr[N])
    mask = np.zeros(max(nwords(nnz), 1), dtype=np.uint64)
    lib().orc_di_graph_edges_free(_d(X), C.c_int64(N), C.c_int32(m), C.c_double(rho), C.c_double(r), _i(colptr), _i(rowval),
                                  _d(lohi), C.c_int32(M), _d(ss_lo), _d(ss_hi), _u(mask))
    return mask[:nwords(nnz)]


def di_fmtstar(X, rho, r, colptr, rowval, nzval, goal_kind, goal, lohi, ss_lo=None, ss_hi=None, init_idx=0, checkpts=True):
    X, N, n = _X(X); m = n // 2
    lohi, M = _boxes(lohi, m); ss_lo = _vec(ss_lo); ss_hi = _vec(ss_hi); goal = _vec(goal)
    colptr = np.ascontiguousarray(colptr, dtype=np.int64); rowval = np.ascontiguousarray(rowval, dtype=np.int64)
    nzval = _vec(nzval)
    A = np.empty(N, dtype=np.int64); Cc = np.empty(N, dtype=np.float64); path = np.empty(N, dtype=np.int64)
    res = FmtResult()
    rc = lib().orc_di_fmtstar(_d(X), C.c_int64(N), C.c_int32(m), C.c_double(rho), C.c_double(r), C.c_int64(init_idx),
                              C.c_int32(int(checkpts)), _i(colptr), _i(rowval), _d(nzval), C.c_int32(goal_kind), _d(goal),
                              _d(lohi), C.c_int32(M), _d(ss_lo), _d(ss_hi), _i(A), _d(Cc), _i(path), C.byref(res))
    return dict(rc=rc, status=int(res.status), cost=float(res.cost), z=int(res.z),
                collision_checks=int(res.collision_checks), A=A, C=Cc, path=path[:res.path_len].copy())


# ---- batch free-space sampler -------------------------------------------------------------------
def philox4x32_10(ctr, key):
    c = (C.c_uint32 * 4)(*ctr); k = (C.c_uint32 * 2)(*key); o = (C.c_uint32 * 4)()
    lib().orc_philox4x32_10(c, k, o)
    return [int(x) for x in o]


def sample_uniforms(seed, c, stream, d):
    u = np.empty(d)
    lib().orc_sample_uniforms(C.c_uint64(seed), C.c_uint64(c), C.c_uint32(stream), C.c_int32(d), _d(u))
    return u


def sample_free(seed, N, d, init, lohi, ss_lo, ss_hi, goal_kind, goal, goal_ct=1, goal_bias=0.0):
    lohi, M = _boxes(lohi, d); ss_lo = _vec(ss_lo); ss_hi = _vec(ss_hi); goal = _vec(goal)
    init = None if init is None else _vec(init)
    W = np.empty((max(N, 0), d)); att = C.c_int64()
    rc = lib().orc_sample_free_biased(C.c_uint64(seed), C.c_int64(N), C.c_int32(d), None if init is None else _d(init), _d(lohi),
                                      C.c_int32(M), _d(ss_lo), _d(ss_hi), C.c_int32(goal_kind), _d(goal), C.c_int32(goal_ct),
                                      C.c_double(goal_bias), _d(W), C.byref(att))
    return rc, W, int(att.value)


# ---- 2-D SAT world ------------------------------------------------------------------------------------
class Shapes2D:
    """Compound2D of Circle / Polygon parts.  shapes: list of ("circle", (cx, cy), r) | ("polygon", [(x, y), ...])."""

    def __init__(self, shapes):
        L = lib()
        L.orc_shape2d_sizeof.restype = C.c_int64
        self.sz = int(L.orc_shape2d_sizeof())
        self.n = len(shapes)
        self.buf = C.create_string_buffer(max(self.n, 1) * self.sz)
        for i, s in enumerate(shapes):
            if s[0] == "circle":
                data = np.array([s[1][0], s[1][1], s[2]], dtype=np.float64); kind, nv = 0, 0
            else:
                data = np.ascontiguousarray(s[1], dtype=np.float64).reshape(-1); kind, nv = 1, len(s[1])
            rc = L.orc_shape2d_build(C.c_int32(kind), C.c_int32(nv), _d(data), C.c_void_p(C.addressof(self.buf) + i * self.sz))
            if rc != 0:
                raise ValueError("shape %d rejected (%d)" % (i, rc))

    @property
    def ptr(self):
        return C.c_void_p(C.addressof(self.buf))


def points_free_2d(P, S, ss_lo=None, ss_hi=None):
    P = np.ascontiguousarray(P, dtype=np.float64); n = len(P)
    mask = np.zeros(max(nwords(n), 1), dtype=np.uint64)
    lib().orc_2d_points_free(_d(P), C.c_int64(n), S.ptr, C.c_int32(S.n), _d(_vec(ss_lo)), _d(_vec(ss_hi)), _u(mask))
    return mask[:nwords(n)]


def motions_free_2d(P, Q, S, ss_lo=None, ss_hi=None):
    P = np.ascontiguousarray(P, dtype=np.float64); Q = np.ascontiguousarray(Q, dtype=np.float64); n = len(P)
    mask = np.zeros(max(nwords(n), 1), dtype=np.uint64)
    lib().orc_2d_motions_free(_d(P), _d(Q), C.c_int64(n), S.ptr, C.c_int32(S.n), _d(_vec(ss_lo)), _d(_vec(ss_hi)), _u(mask))
    return mask[:nwords(n)]


def graph_edges_free_2d(X, colptr, rowval, S, ss_lo=None, ss_hi=None):
    X, N, d = _X(X)
    colptr = np.ascontiguousarray(colptr, dtype=np.int64); rowval = np.ascontiguousarray(rowval, dtype=np.int64)
    nnz = int(colptr[-1])
    mask = np.zeros(max(nwords(nnz), 1), dtype=np.uint64)
    lib().orc_2d_graph_edges_free(_d(X), C.c_int64(N), _i(colptr), _i(rowval), S.ptr, C.c_int32(S.n), _d(_vec(ss_lo)), _d(_vec(ss_hi)), _u(mask))
    return mask[:nwords(nnz)]


# ---- Dubins car ------------------------------------------------------------------------------------------
def dubins(s1, s2, rt=1.0, sp=1.0):
    """(cost, controls[3][3] = (t, speed, curvature)) of simplecars.jl:198-215."""
    s1 = _vec(s1); s2 = _vec(s2)
    path = np.zeros((3, 3))
    L = lib(); L.orc_dubins.restype = C.c_double
    c = L.orc_dubins(_d(s1), _d(s2), C.c_double(rt), C.c_double(sp), _d(path))
    return float(c), path


def dubins_waypoints(v, w, rt=1.0, sp=1.0):
    v = _vec(v); w = _vec(w)
    wps = np.zeros((160, 3))
    n = lib().orc_dubins_waypoints(_d(v), _d(w), C.c_double(rt), C.c_double(sp), _d(wps))
    return wps[:n].copy()


def dubins_is_free_motion(v, w, rt, sp, lohi, ss_lo, ss_hi):
    v = _vec(v); w = _vec(w); lohi, M = _boxes(lohi, 2); ns = C.c_int32()
    ok = lib().orc_dubins_is_free_motion(_d(v), _d(w), C.c_double(rt), C.c_double(sp), _d(lohi), C.c_int32(M), _d(_vec(ss_lo)),
                                         _d(_vec(ss_hi)), C.byref(ns))
    return bool(ok), int(ns.value)


def dubins_graph(X, rt, sp, r):
    """Backward sets as 0-based CSC (colptr, rowval, nzval): column j = sources i with dubins(i -> j) <= r."""
    X, N, d = _X(X)
    L = lib(); L.orc_dubins_graph.restype = C.c_int64
    colptr = np.zeros(N + 1, dtype=np.int64)
    nnz = int(L.orc_dubins_graph(_d(X), C.c_int64(N), C.c_double(rt), C.c_double(sp), C.c_double(r), _i(colptr), None, None))
    rowval = np.zeros(max(nnz, 1), dtype=np.int64); nzval = np.zeros(max(nnz, 1))
    L.orc_dubins_graph(_d(X), C.c_int64(N), C.c_double(rt), C.c_double(sp), C.c_double(r), None, _i(rowval), _d(nzval))
    return colptr, rowval[:nnz], nzval[:nnz]


def dubins_graph_edges_free(X, rt, sp, colptr, rowval, lohi, ss_lo, ss_hi):
    X, N, d = _X(X); lohi, M = _boxes(lohi, 2)
    colptr = np.ascontiguousarray(colptr, dtype=np.int64); rowval = np.ascontiguousarray(rowval, dtype=np.int64)
    nnz = int(colptr[-1])
    mask = np.zeros(max(nwords(nnz), 1), dtype=np.uint64); nseg = np.zeros(max(nnz, 1), dtype=np.uint8)
    lib().orc_dubins_graph_edges_free(_d(X), C.c_int64(N), C.c_double(rt), C.c_double(sp), _i(colptr), _i(rowval), _d(lohi), C.c_int32(M),
                                      _d(_vec(ss_lo)), _d(_vec(ss_hi)), _u(mask), nseg.ctypes.data_as(C.POINTER(C.c_uint8)))
    return mask[:nwords(nnz)], nseg[:nnz]


def dubins_fmtstar(X, rt, sp, colptr, rowval, nzval, goal_kind, goal, lohi, ss_lo, ss_hi, init_idx=0, checkpts=True):
    X, N, d = _X(X); lohi, M = _boxes(lohi, 2); goal = _vec(goal)
    colptr = np.ascontiguousarray(colptr, dtype=np.int64); rowval = np.ascontiguousarray(rowval, dtype=np.int64); nzval = _vec(nzval)
    A = np.empty(N, dtype=np.int64); Cc = np.empty(N, dtype=np.float64); path = np.empty(N, dtype=np.int64)
    res = FmtResult()
    rc = lib().orc_dubins_fmtstar(_d(X), C.c_int64(N), C.c_double(rt), C.c_double(sp), C.c_int64(init_idx), C.c_int32(int(checkpts)),
                                  _i(colptr), _i(rowval), _d(nzval), C.c_int32(goal_kind), _d(goal), _d(lohi), C.c_int32(M),
                                  _d(_vec(ss_lo)), _d(_vec(ss_hi)), _i(A), _d(Cc), _i(path), C.byref(res))
    return dict(rc=rc, status=int(res.status), cost=float(res.cost), z=int(res.z), collision_checks=int(res.collision_checks),
                A=A, C=Cc, path=path[:res.path_len].copy())


# ---- Monte-Carlo collision probability per edge (BASELINE configs[4]) ------------------------------------------------
def mc_edges(X, src, dst, sigma, rollouts, seed, lohi, ss_lo=None, ss_hi=None):
    X, N, d = _X(X); lohi, M = _boxes(lohi, d)
    src = np.ascontiguousarray(src, dtype=np.int64); dst = np.ascontiguousarray(dst, dtype=np.int64)
    hits = np.zeros(max(len(src), 1), dtype=np.int64)
    lib().orc_mc_edges(_d(X), C.c_int32(d), _i(src), _i(dst), C.c_int64(len(src)), C.c_double(sigma), C.c_int64(rollouts),
                       C.c_uint64(seed), _d(lohi), C.c_int32(M), _d(_vec(ss_lo)), _d(_vec(ss_hi)), _i(hits))
    return hits[:len(src)]


def mc_is_edges(X, src, dst, sigma, rollouts, seed, lohi, ss_lo=None, ss_hi=None):
    """Importance-sampling estimator of the same probability: sum of the colliding rollouts' weights, quantised to 2^-40 (uint64)."""
    X, N, d = _X(X); lohi, M = _boxes(lohi, d)
    src = np.ascontiguousarray(src, dtype=np.int64); dst = np.ascontiguousarray(dst, dtype=np.int64)
    wsum = np.zeros(max(len(src), 1), dtype=np.uint64)
    lib().orc_mc_is_edges(_d(X), C.c_int32(d), _i(src), _i(dst), C.c_int64(len(src)), C.c_double(sigma), C.c_int64(rollouts),
                          C.c_uint64(seed), _d(lohi), C.c_int32(M), _d(_vec(ss_lo)), _d(_vec(ss_hi)),
                          wsum.ctypes.data_as(C.POINTER(C.c_uint64)))
    return wsum[:len(src)]


def mc_ais_edges(X, src, dst, sigma, rollouts, seed, lohi, ss_lo=None, ss_hi=None):
    """ADAPTIVE importance sampling (pilot -> cross-entropy mean shift -> mixture): (wsum uint64 2^-40, shifts (E, 2 d) in noise units)."""
    X, N, d = _X(X); lohi, M = _boxes(lohi, d)
    src = np.ascontiguousarray(src, dtype=np.int64); dst = np.ascontiguousarray(dst, dtype=np.int64)
    wsum = np.zeros(max(len(src), 1), dtype=np.uint64)
    sh = np.zeros((max(len(src), 1), 2 * d))
    lib().orc_mc_ais_edges(_d(X), C.c_int32(d), _i(src), _i(dst), C.c_int64(len(src)), C.c_double(sigma), C.c_int64(rollouts),
                           C.c_uint64(seed), _d(lohi), C.c_int32(M), _d(_vec(ss_lo)), _d(_vec(ss_hi)),
                           wsum.ctypes.data_as(C.POINTER(C.c_uint64)), _d(sh))
    return wsum[:len(src)], sh[:len(src)]


# ---- Reeds-Shepp car (second half) ---------------------------------------------------------------------
def reedsshepp(s1, s2, rt=1.0, sp=1.0):
    """(cost, controls[L][3] = (t, speed, curvature)) of simplecars.jl:265-363."""
    s1 = _vec(s1); s2 = _vec(s2)
    path = np.zeros((5, 3)); L = C.c_int32()
    Lb = lib(); Lb.orc_reedsshepp.restype = C.c_double
    c = Lb.orc_reedsshepp(_d(s1), _d(s2), C.c_double(rt), C.c_double(sp), _d(path), C.byref(L))
    return float(c), path[:L.value].copy()


def car_waypoints(kind, v, w, rt=1.0, sp=1.0):
    v = _vec(v); w = _vec(w)
    wps = np.zeros((160, 3))
    n = lib().orc_car_waypoints(C.c_int32(kind), _d(v), _d(w), C.c_double(rt), C.c_double(sp), _d(wps))
    return wps[:n].copy()


def car_is_free_motion(kind, v, w, rt, sp, lohi, ss_lo, ss_hi):
    v = _vec(v); w = _vec(w); lohi, M = _boxes(lohi, 2); ns = C.c_int32()
    ok = lib().orc_car_is_free_motion(C.c_int32(kind), _d(v), _d(w), C.c_double(rt), C.c_double(sp), _d(lohi), C.c_int32(M),
                                      _d(_vec(ss_lo)), _d(_vec(ss_hi)), C.byref(ns))
    return bool(ok), int(ns.value)


def rs_graph(X, rt, sp, r):
    """inball sets of the Reeds-Shepp space as 0-based CSC: column v = { w : rs(v -> w) <= r }, nzval = rs(v -> w)."""
    X, N, d = _X(X)
    L = lib(); L.orc_rs_graph.restype = C.c_int64
    colptr = np.zeros(N + 1, dtype=np.int64)
    nnz = int(L.orc_rs_graph(_d(X), C.c_int64(N), C.c_double(rt), C.c_double(sp), C.c_double(r), _i(colptr), None, None))
    rowval = np.zeros(max(nnz, 1), dtype=np.int64); nzval = np.zeros(max(nnz, 1))
    L.orc_rs_graph(_d(X), C.c_int64(N), C.c_double(rt), C.c_double(sp), C.c_double(r), None, _i(rowval), _d(nzval))
    return colptr, rowval[:nnz], nzval[:nnz]


def car_graph_edges_free(kind, X, rt, sp, colptr, rowval, lohi, ss_lo, ss_hi):
    X, N, d = _X(X); lohi, M = _boxes(lohi, 2)
    colptr = np.ascontiguousarray(colptr, dtype=np.int64); rowval = np.ascontiguousarray(rowval, dtype=np.int64)
    nnz = int(colptr[-1])
    mask = np.zeros(max(nwords(nnz), 1), dtype=np.uint64); nseg = np.zeros(max(nnz, 1), dtype=np.uint8)
    lib().orc_car_graph_edges_free(C.c_int32(kind), _d(X), C.c_int64(N), C.c_double(rt), C.c_double(sp), _i(colptr), _i(rowval), _d(lohi),
                                   C.c_int32(M), _d(_vec(ss_lo)), _d(_vec(ss_hi)), _u(mask), nseg.ctypes.data_as(C.POINTER(C.c_uint8)))
    return mask[:nwords(nnz)], nseg[:nnz]


def rs_fmtstar(X, rt, sp, colptr, rowval, nzval, goal_kind, goal, lohi, ss_lo, ss_hi, init_idx=0, checkpts=True):
    X, N, d = _X(X); lohi, M = _boxes(lohi, 2); goal = _vec(goal)
    colptr = np.ascontiguousarray(colptr, dtype=np.int64); rowval = np.ascontiguousarray(rowval, dtype=np.int64); nzval = _vec(nzval)
    A = np.empty(N, dtype=np.int64); Cc = np.empty(N, dtype=np.float64); path = np.empty(N, dtype=np.int64)
    res = FmtResult()
    rc = lib().orc_rs_fmtstar(_d(X), C.c_int64(N), C.c_double(rt), C.c_double(sp), C.c_int64(init_idx), C.c_int32(int(checkpts)),
                              _i(colptr), _i(rowval), _d(nzval), C.c_int32(goal_kind), _d(goal), _d(lohi), C.c_int32(M),
                              _d(_vec(ss_lo)), _d(_vec(ss_hi)), _i(A), _d(Cc), _i(path), C.byref(res))
    return dict(rc=rc, status=int(res.status), cost=float(res.cost), z=int(res.z), collision_checks=int(res.collision_checks),
                A=A, C=Cc, path=path[:res.path_len].copy())


# ---- closest obstacle points in a Mahalanobis metric --------------------------------------------------
def bvls(A, b, l, u):
    """bvls(A, b, l, u) of bvls.jl:19-218 (A square); returns (x, iterations) -- iterations < 0: ran out (`nothing`)."""
    A = np.ascontiguousarray(A, dtype=np.float64); n = A.shape[0]
    b = _vec(b); l = _vec(l); u = _vec(u); x = np.zeros(n)
    it = lib().orc_bvls(C.c_int32(n), _d(A), _d(b), _d(l), _d(u), _d(x))
    return x, int(it)


def _W(W, d):
    W = np.ascontiguousarray(W, dtype=np.float64)
    assert W.shape == (d, d)
    return W


def closest_boxes(P, lohi, W):
    """closest(p, BL, W) (boxesND.jl:72-81) per point: (d2min, vmin, kmin 0-based / -1, failures).  failures = (point, box)
    pairs on which bvls ran out of its 10n iterations (the reference then throws); they are skipped in the minimum."""
    P = np.ascontiguousarray(np.atleast_2d(P), dtype=np.float64); n, d = P.shape
    lohi, M = _boxes(lohi, d); W = _W(W, d)
    d2 = np.zeros(max(n, 1)); v = np.zeros((max(n, 1), d)); k = np.zeros(max(n, 1), dtype=np.int64)
    L = lib(); L.orc_closest_boxes.restype = C.c_int64
    bad = L.orc_closest_boxes(_d(P), C.c_int64(n), _d(lohi), C.c_int32(M), _d(W), C.c_int32(d), _d(d2), _d(v), _i(k))
    return d2[:n], v[:n], k[:n], int(bad)


def closeR_boxes(P, lohi, W, r2):
    """closeR(p, BL, W, r2) (boxesND.jl:83-86) per point as (ptr, idx 0-based, d2, v)."""
    P = np.ascontiguousarray(np.atleast_2d(P), dtype=np.float64); n, d = P.shape
    lohi, M = _boxes(lohi, d); W = _W(W, d)
    L = lib(); L.orc_closeR_boxes.restype = C.c_int64
    ptr = np.zeros(n + 1, dtype=np.int64)
    tot = int(L.orc_closeR_boxes(_d(P), C.c_int64(n), _d(lohi), C.c_int32(M), _d(W), C.c_int32(d), C.c_double(r2), _i(ptr), None, None, None))
    idx = np.zeros(max(tot, 1), dtype=np.int64); d2 = np.zeros(max(tot, 1)); v = np.zeros((max(tot, 1), d))
    L.orc_closeR_boxes(_d(P), C.c_int64(n), _d(lohi), C.c_int32(M), _d(W), C.c_int32(d), C.c_double(r2), _i(ptr), _i(idx), _d(d2), _d(v))
    return ptr, idx[:tot], d2[:tot], v[:tot]


def closest_shapes(P, S, W=None):
    """closest(p, C::Compound2D [, W]) (SAT2D.jl:260-279) per point."""
    P = np.ascontiguousarray(np.atleast_2d(P), dtype=np.float64); n = len(P)
    Wp = None if W is None else _d(_W(W, 2))
    d2 = np.zeros(max(n, 1)); v = np.zeros((max(n, 1), 2)); k = np.zeros(max(n, 1), dtype=np.int64)
    L = lib(); L.orc_closest_shapes.restype = C.c_int64
    bad = L.orc_closest_shapes(_d(P), C.c_int64(n), S.ptr, C.c_int32(S.n), Wp, _d(d2), _d(v), _i(k))
    return d2[:n], v[:n], k[:n], int(bad)


def closeR_shapes(P, S, W, r2):
    P = np.ascontiguousarray(np.atleast_2d(P), dtype=np.float64); n = len(P)
    W = _W(W, 2)
    L = lib(); L.orc_closeR_shapes.restype = C.c_int64
    ptr = np.zeros(n + 1, dtype=np.int64)
    tot = int(L.orc_closeR_shapes(_d(P), C.c_int64(n), S.ptr, C.c_int32(S.n), _d(W), C.c_double(r2), _i(ptr), None, None, None))
    idx = np.zeros(max(tot, 1), dtype=np.int64); d2 = np.zeros(max(tot, 1)); v = np.zeros((max(tot, 1), 2))
    L.orc_closeR_shapes(_d(P), C.c_int64(n), S.ptr, C.c_int32(S.n), _d(W), C.c_double(r2), _i(ptr), _i(idx), _d(d2), _d(v))
    return ptr, idx[:tot], d2[:tot], v[:tot]
