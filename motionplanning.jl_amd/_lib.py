"""ctypes loader for libmpfmt.so (the C ABI of include/mpfmt.h).

The product path has NO CPU fallback: if the shared library is missing this module raises, and if
no gfx950 device is visible `Context()` raises `MPFMTError` (MPFMT_ERR_NODEVICE).
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.environ.get("MPFMT_LIB_PATH") or os.path.join(_HERE, "libmpfmt.so")   # override: kernel experiments (tools/)
_LIB = None

OK, ERR_ARG, ERR_STATE, ERR_HIP, ERR_NODEVICE, ERR_CAPACITY, ERR_INFEASIBLE = 0, -1, -2, -3, -4, -5, -6
RETRY = 1            # mpfmt_allgather_free_mask_finish on a ctx driven inside a group: relaunch every ctx, then finish again
GOAL_RECT, GOAL_BALL, GOAL_POINT = 0, 1, 2
MAX_DIM = 16

c_d_p = C.POINTER(C.c_double)
c_i64_p = C.POINTER(C.c_int64)
c_u64_p = C.POINTER(C.c_uint64)
c_u8_p = C.POINTER(C.c_uint8)


class MPFMTError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("libmpfmt error %d: %s" % (code, msg))
        self.code = code


class FmtResult(C.Structure):
    _fields_ = [("status", C.c_int32), ("cost", C.c_double), ("z", C.c_int64), ("collision_checks", C.c_int64),
                ("path_len", C.c_int64), ("nnz", C.c_int64), ("ms_graph", C.c_double), ("ms_sweep", C.c_double),
                ("ms_host_loop", C.c_double)]


class WfInfo(C.Structure):
    _fields_ = [("done", C.c_int32), ("nz", C.c_int32), ("nx", C.c_int32), ("nconn", C.c_int32), ("ntrip", C.c_int32),
                ("iters", C.c_int64), ("checks", C.c_int64), ("cmin", C.c_double), ("tot_z", C.c_int64), ("tot_x", C.c_int64),
                ("tot_conn", C.c_int64)]


WF_SINGLE, WF_EAGER, WF_LAZY = 1, 2, 4
COMM_ID_BYTES = 128

# every symbol include/mpfmt.h declares: (name, restype, argtypes)
SYMBOLS = [
    ("mpfmt_ctx_create", C.c_int32, [C.c_int32, C.POINTER(C.c_void_p)]),
    ("mpfmt_ctx_destroy", C.c_int32, [C.c_void_p]),
    ("mpfmt_last_error", C.c_char_p, [C.c_void_p]),
    ("mpfmt_version", C.c_char_p, []),
    ("mpfmt_set_stream", C.c_int32, [C.c_void_p, C.c_void_p]),
    ("mpfmt_set_shard", C.c_int32, [C.c_void_p, C.c_int32, C.c_int32]),
    ("mpfmt_upload_samples", C.c_int32, [C.c_void_p, c_d_p, C.c_int64, C.c_int32]),
    ("mpfmt_upload_samples_device", C.c_int32, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int32]),
    ("mpfmt_upload_boxes", C.c_int32, [C.c_void_p, c_d_p, C.c_int32, C.c_int32, c_d_p, c_d_p, C.c_int32]),
    ("mpfmt_rdisc_count", C.c_int32, [C.c_void_p, C.c_double, c_i64_p, c_i64_p]),
    ("mpfmt_rdisc_fill", C.c_int32, [C.c_void_p, c_i64_p, c_d_p]),
    ("mpfmt_rdisc_query", C.c_int32, [C.c_void_p, C.c_int64, C.c_double, c_i64_p, c_d_p, C.c_int64, c_i64_p]),
    ("mpfmt_points_free", C.c_int32, [C.c_void_p, c_i64_p, C.c_int64, c_u64_p]),
    ("mpfmt_edges_free", C.c_int32, [C.c_void_p, c_i64_p, c_i64_p, C.c_int64, c_u64_p]),
    ("mpfmt_graph_edges_free", C.c_int32, [C.c_void_p, c_u64_p]),
    ("mpfmt_states_free", C.c_int32, [C.c_void_p, c_d_p, C.c_int64, c_u64_p]),
    ("mpfmt_motions_free", C.c_int32, [C.c_void_p, c_d_p, c_d_p, C.c_int64, c_u64_p]),
    ("mpfmt_euclid_steer", C.c_int32, [C.c_void_p, c_i64_p, c_i64_p, C.c_int64, c_d_p, c_d_p]),
    ("mpfmt_euclid_propagate", C.c_int32, [C.c_void_p, c_i64_p, C.c_int64, c_d_p, c_d_p, c_d_p, c_d_p]),
    ("mpfmt_expand", C.c_int32, [C.c_void_p, c_u64_p, c_u64_p, c_u64_p, c_d_p, c_i64_p, C.c_int64,
                                 c_i64_p, c_i64_p, c_d_p, c_u8_p, C.c_int64, c_i64_p]),
    ("mpfmt_fmtstar", C.c_int32, [C.c_void_p, C.c_double, C.c_int64, C.c_int32, C.c_int32, c_d_p,
                                  c_i64_p, c_d_p, c_i64_p, C.POINTER(FmtResult)]),
    ("mpfmt_host_fmt_recursion", C.c_int32, [C.c_int64, C.c_int32, c_d_p, c_i64_p, C.POINTER(C.c_int32), c_d_p, c_u64_p, c_u64_p,
                                             c_d_p, c_d_p, C.c_int64, C.c_int32, c_d_p, c_i64_p, c_d_p, c_i64_p,
                                             C.POINTER(FmtResult)]),
    ("mpfmt_mc_edges_collision", C.c_int32, [C.c_void_p, c_i64_p, c_i64_p, C.c_int64, C.c_double, C.c_int64, C.c_uint64, c_i64_p]),
    ("mpfmt_mc_edges_collision_is", C.c_int32, [C.c_void_p, c_i64_p, c_i64_p, C.c_int64, C.c_double, C.c_int64, C.c_uint64, C.POINTER(C.c_uint64)]),
    ("mpfmt_mc_edges_collision_ais", C.c_int32, [C.c_void_p, c_i64_p, c_i64_p, C.c_int64, C.c_double, C.c_int64, C.c_uint64, C.POINTER(C.c_uint64),
                                                 c_d_p]),
    ("mpfmt_dubins_graph_count", C.c_int32, [C.c_void_p, C.c_double, C.c_double, C.c_double, c_i64_p, c_i64_p]),
    ("mpfmt_dubins_graph_fill", C.c_int32, [C.c_void_p, c_i64_p, c_d_p]),
    ("mpfmt_dubins_graph_edges_free", C.c_int32, [C.c_void_p, c_u64_p, c_u8_p]),
    ("mpfmt_dubins_steer", C.c_int32, [C.c_void_p, c_d_p, c_d_p, C.c_int64, C.c_double, C.c_double, c_d_p, c_d_p]),
    ("mpfmt_dubins_fmtstar", C.c_int32, [C.c_void_p, C.c_double, C.c_double, C.c_double, C.c_int64, C.c_int32, C.c_int32, c_d_p,
                                         c_i64_p, c_d_p, c_i64_p, C.POINTER(FmtResult)]),
    ("mpfmt_reedsshepp_graph_count", C.c_int32, [C.c_void_p, C.c_double, C.c_double, C.c_double, c_i64_p, c_i64_p]),
    ("mpfmt_reedsshepp_graph_fill", C.c_int32, [C.c_void_p, c_i64_p, c_d_p]),
    ("mpfmt_reedsshepp_graph_edges_free", C.c_int32, [C.c_void_p, c_u64_p, c_u8_p]),
    ("mpfmt_reedsshepp_steer", C.c_int32, [C.c_void_p, c_d_p, c_d_p, C.c_int64, C.c_double, C.c_double, c_d_p, c_d_p,
                                           C.POINTER(C.c_int32)]),
    ("mpfmt_reedsshepp_fmtstar", C.c_int32, [C.c_void_p, C.c_double, C.c_double, C.c_double, C.c_int64, C.c_int32, C.c_int32, c_d_p,
                                             c_i64_p, c_d_p, c_i64_p, C.POINTER(FmtResult)]),
    ("mpfmt_closest", C.c_int32, [C.c_void_p, c_d_p, C.c_int64, c_d_p, c_d_p, c_d_p, c_i64_p, c_i64_p]),
    ("mpfmt_closeR", C.c_int32, [C.c_void_p, c_d_p, C.c_int64, c_d_p, C.c_double, c_i64_p, C.c_int64, c_i64_p, c_d_p, c_d_p,
                                 c_i64_p, c_i64_p]),
    ("mpfmt_upload_shapes2d", C.c_int32, [C.c_void_p, C.c_int32, C.POINTER(C.c_int32), C.POINTER(C.c_int32), c_d_p, c_d_p, c_d_p]),
    ("mpfmt_graph_import", C.c_int32, [C.c_void_p, C.c_double, c_i64_p, c_i64_p, c_d_p]),
    ("mpfmt_sample_free", C.c_int32, [C.c_void_p, C.c_uint64, C.c_int64, c_d_p, C.c_int32, c_d_p, C.c_int32, c_d_p, c_i64_p]),
    ("mpfmt_sample_free_biased", C.c_int32, [C.c_void_p, C.c_uint64, C.c_int64, c_d_p, C.c_int32, c_d_p, C.c_int32, C.c_double, c_d_p,
                                             c_i64_p]),
    ("mpfmt_path_free", C.c_int32, [C.c_void_p, c_d_p, C.c_int64, C.POINTER(C.c_int32), c_u64_p]),
    ("mpfmt_di_graph_count", C.c_int32, [C.c_void_p, C.c_double, C.c_double, c_i64_p, c_i64_p]),
    ("mpfmt_di_graph_fill", C.c_int32, [C.c_void_p, c_i64_p, c_d_p, c_d_p]),
    ("mpfmt_di_graph_edges_free", C.c_int32, [C.c_void_p, c_u64_p, c_u8_p]),
    ("mpfmt_di_graph_step_device", C.c_int32, [C.c_void_p, C.c_double, C.c_double, c_i64_p]),
    ("mpfmt_di_graph_device_ptrs", C.c_int32, [C.c_void_p] + [C.POINTER(C.c_void_p)] * 6),
    ("mpfmt_di_steer", C.c_int32, [C.c_void_p, c_d_p, c_d_p, C.c_int64, C.c_int32, C.c_double, C.c_double, c_d_p, c_d_p]),
    ("mpfmt_di_fmtstar", C.c_int32, [C.c_void_p, C.c_double, C.c_double, C.c_int64, C.c_int32, C.c_int32, c_d_p,
                                     c_i64_p, c_d_p, c_i64_p, C.POINTER(FmtResult)]),
    ("mpfmt_dubins_fmtstar_wavefront", C.c_int32, [C.c_void_p, C.c_double, C.c_double, C.c_double, C.c_int64, C.c_int32, C.c_int32, c_d_p,
                                                   C.c_double, C.c_int32, c_i64_p, c_d_p, c_i64_p, C.POINTER(FmtResult), C.POINTER(WfInfo)]),
    ("mpfmt_reedsshepp_fmtstar_wavefront", C.c_int32, [C.c_void_p, C.c_double, C.c_double, C.c_double, C.c_int64, C.c_int32, C.c_int32, c_d_p,
                                                       C.c_double, C.c_int32, c_i64_p, c_d_p, c_i64_p, C.POINTER(FmtResult), C.POINTER(WfInfo)]),
    ("mpfmt_di_fmtstar_wavefront", C.c_int32, [C.c_void_p, C.c_double, C.c_double, C.c_int64, C.c_int32, C.c_int32, c_d_p, C.c_double,
                                               C.c_int32, c_i64_p, c_d_p, c_i64_p, C.POINTER(FmtResult), C.POINTER(WfInfo)]),
    ("mpfmt_graph_build_device", C.c_int32, [C.c_void_p, C.c_double, c_i64_p]),
    ("mpfmt_graph_step_device", C.c_int32, [C.c_void_p, C.c_double, c_i64_p]),
    ("mpfmt_graph_step_launch", C.c_int32, [C.c_void_p, C.c_double]),
    ("mpfmt_graph_step_finish", C.c_int32, [C.c_void_p, c_i64_p]),
    ("mpfmt_graph_sweep_device", C.c_int32, [C.c_void_p]),
    ("mpfmt_graph_device_ptrs", C.c_int32, [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p),
                                            C.POINTER(C.c_void_p), C.POINTER(C.c_void_p)]),
    ("mpfmt_graph_export", C.c_int32, [C.c_void_p, c_i64_p, c_i64_p, c_d_p, c_u64_p, c_d_p]),
    ("mpfmt_pinned_alloc", C.c_int32, [C.c_int64, C.POINTER(C.c_void_p)]),
    ("mpfmt_pinned_free", C.c_int32, [C.c_void_p]),
    ("mpfmt_export_arena", C.c_int32, [C.c_void_p, C.c_int64, C.POINTER(C.c_void_p)]),
    ("mpfmt_graph_export_pinned", C.c_int32, [C.c_void_p, C.POINTER(c_i64_p), C.POINTER(c_i64_p), C.POINTER(c_d_p), C.POINTER(c_u64_p),
                                              c_i64_p, C.POINTER(C.c_double)]),
    ("mpfmt_rdisc_stream", C.c_int32, [C.c_void_p, C.c_double, c_d_p, c_u64_p, C.c_int32, c_i64_p, c_i64_p, c_i64_p, c_d_p, c_i64_p]),
    ("mpfmt_shard_info", C.c_int32, [C.c_void_p, c_i64_p, c_i64_p, c_i64_p]),
    ("mpfmt_fmtstar_wavefront", C.c_int32, [C.c_void_p, C.c_double, C.c_int64, C.c_int32, C.c_int32, c_d_p, C.c_double, C.c_int32,
                                            c_i64_p, c_d_p, c_i64_p, C.POINTER(FmtResult), C.POINTER(WfInfo)]),
    ("mpfmt_wf_begin", C.c_int32, [C.c_void_p, C.c_double, C.c_int64, C.c_int32, C.c_int32, c_d_p, C.c_double, C.c_int32]),
    ("mpfmt_wf_step", C.c_int32, [C.c_void_p, C.POINTER(WfInfo)]),
    ("mpfmt_wf_state", C.c_int32, [C.c_void_p, c_u64_p, c_u64_p, c_d_p, c_i64_p]),
    ("mpfmt_wf_batch", C.c_int32, [C.c_void_p, c_i64_p, C.c_int64, c_i64_p]),
    ("mpfmt_wf_triples", C.c_int32, [C.c_void_p, C.c_int64, c_i64_p, c_i64_p, c_d_p, c_i64_p]),
    ("mpfmt_wf_commit", C.c_int32, [C.c_void_p, C.c_int64, c_i64_p, c_i64_p, c_d_p]),
    ("mpfmt_wf_finish", C.c_int32, [C.c_void_p, c_i64_p, c_d_p, c_i64_p, C.POINTER(FmtResult)]),
    ("mpfmt_comm_unique_id", C.c_int32, [c_u8_p]),
    ("mpfmt_comm_create", C.c_int32, [C.c_void_p, C.c_int32, C.c_int32, c_u8_p]),
    ("mpfmt_comm_destroy", C.c_int32, [C.c_void_p]),
    ("mpfmt_group_begin", C.c_int32, []),
    ("mpfmt_group_end", C.c_int32, []),
    ("mpfmt_allgather_free_mask", C.c_int32, [C.c_void_p, C.POINTER(C.c_void_p), c_i64_p, c_i64_p, c_i64_p]),
    ("mpfmt_allgather_free_mask_launch", C.c_int32, [C.c_void_p, C.c_int64]),
    ("mpfmt_allgather_free_mask_finish", C.c_int32, [C.c_void_p, C.POINTER(C.c_void_p), c_i64_p, c_i64_p, c_i64_p]),
    ("mpfmt_allgather_free_mask_relaunch", C.c_int32, [C.c_void_p]),
    ("mpfmt_set_state_bounds", C.c_int32, [C.c_void_p, c_d_p, c_d_p, C.c_int32]),
    ("mpfmt_timing_reset", C.c_int32, [C.c_void_p]),
    ("mpfmt_timing_get", C.c_int32, [C.c_void_p, C.c_char_p, c_d_p, c_i64_p]),
    ("mpfmt_set_option", C.c_int32, [C.c_void_p, C.c_char_p, C.c_int64]),
    ("mpfmt_get_stat", C.c_int32, [C.c_void_p, C.c_char_p, c_i64_p]),
    ("mpfmt_graph_stats", C.c_int32, [C.c_void_p, c_i64_p, c_i64_p, c_i64_p, c_i64_p]),
]


def so_path():
    return _SO


def lib():
    """Load libmpfmt.so.  torch is imported first so that the one HIP runtime already mapped into the
    process (torch bundles libamdhip64.so.7) also serves this library (same SONAME)."""
    global _LIB
    if _LIB is None:
        if not os.path.exists(_SO):
            raise ImportError("libmpfmt.so is not built (%s); run `python __graft_entry__.py` / `make -C %s`; "
                              "there is no CPU fallback" % (_SO, os.path.join(_HERE, "csrc")))
        try:
            import torch  # noqa: F401  (maps the HIP runtime)
        except Exception:  # pragma: no cover - torch is plumbing only
            pass
        L = C.CDLL(_SO)
        for name, res, args in SYMBOLS:
            f = getattr(L, name)
            f.restype = res
            f.argtypes = args
        _LIB = L
    return _LIB


def comm_unique_id():
    """128-byte RCCL id (rank 0 calls this; the host distributes it)."""
    u = np.zeros(COMM_ID_BYTES, dtype=np.uint8)
    rc = lib().mpfmt_comm_unique_id(u.ctypes.data_as(c_u8_p))
    if rc != OK:
        raise MPFMTError(rc, lib().mpfmt_last_error(None).decode())
    return u.tobytes()


def group_begin():
    """ncclGroupStart: one host thread driving several ctxs brackets every round of *_launch calls (include/mpfmt.h)."""
    rc = lib().mpfmt_group_begin()
    if rc != OK:
        raise MPFMTError(rc, lib().mpfmt_last_error(None).decode())


def group_end():
    rc = lib().mpfmt_group_end()
    if rc != OK:
        raise MPFMTError(rc, lib().mpfmt_last_error(None).decode())


def nwords(n):
    return (int(n) + 63) // 64


def unpack_bits(mask, n):
    b = np.unpackbits(np.ascontiguousarray(mask).view(np.uint8), bitorder="little")
    return b[:n].astype(bool)


def pack_bits(bits):
    bits = np.asarray(bits, dtype=bool)
    pad = np.zeros(max(nwords(bits.size), 1) * 64, dtype=np.uint8)
    pad[:bits.size] = bits
    return np.packbits(pad, bitorder="little").view(np.uint64)


def _dp(a):
    return None if a is None else a.ctypes.data_as(c_d_p)


def _ip(a):
    return None if a is None else a.ctypes.data_as(c_i64_p)


def _dp_or_none(a):
    return None if a is None else a.ctypes.data_as(c_d_p)


def _up(a):
    return None if a is None else a.ctypes.data_as(c_u64_p)


def host_fmt_recursion(X, colptr0, rowval0, nzval, efree, F, goal_kind, goal_params, ss_lo=None, ss_hi=None, init_idx=1):
    """The sequential part of fmtstar! (fmt.jl:43-101) on a finished graph; host only, no GPU needed.
    colptr0 / rowval0 are 0-based (rowval0 int32), efree / F packed uint64 bit masks (F None: checkpts=false)."""
    X = np.ascontiguousarray(X, dtype=np.float64)
    N, d = X.shape
    colptr0 = np.ascontiguousarray(colptr0, dtype=np.int64)
    rowval0 = np.ascontiguousarray(rowval0, dtype=np.int32)
    nzval = np.ascontiguousarray(nzval, dtype=np.float64)
    efree = np.ascontiguousarray(efree, dtype=np.uint64)
    Fp = None if F is None else np.ascontiguousarray(F, dtype=np.uint64)
    lo = None if ss_lo is None else np.ascontiguousarray(ss_lo, dtype=np.float64)
    hi = None if ss_hi is None else np.ascontiguousarray(ss_hi, dtype=np.float64)
    g = np.ascontiguousarray(goal_params, dtype=np.float64)
    A = np.empty(N, dtype=np.int64); Cc = np.empty(N, dtype=np.float64); path = np.empty(N, dtype=np.int64)
    res = FmtResult()
    rc = lib().mpfmt_host_fmt_recursion(N, d, _dp(X), _ip(colptr0), rowval0.ctypes.data_as(C.POINTER(C.c_int32)), _dp(nzval),
                                        _up(efree), None if Fp is None else _up(Fp), None if lo is None else _dp(lo),
                                        None if hi is None else _dp(hi), int(init_idx), int(goal_kind), _dp(g),
                                        _ip(A), _dp(Cc), _ip(path), C.byref(res))
    if rc != 0:
        raise MPFMTError(rc, "mpfmt_host_fmt_recursion rejected its arguments")
    return dict(status=int(res.status), cost=float(res.cost), z=int(res.z), collision_checks=int(res.collision_checks),
                nnz=int(res.nnz), ms_host_loop=float(res.ms_host_loop), A=A, C=Cc, path=path[:res.path_len].copy())


class Context:
    """One GPU context = one `mpfmt_ctx`.  Thin, numpy-in / numpy-out; indices 1-based like the ABI."""

    def __init__(self, device=0):
        self._L = lib()
        h = C.c_void_p()
        rc = self._L.mpfmt_ctx_create(int(device), C.byref(h))
        if rc != OK:
            raise MPFMTError(rc, self._L.mpfmt_last_error(None).decode())
        self._h = h
        self.N = 0
        self.d = 0
        self.dw = 0
        self.nnz = None

    def close(self):
        if getattr(self, "_h", None):
            self._L.mpfmt_ctx_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def _chk(self, rc):
        if rc != OK:
            raise MPFMTError(rc, self._L.mpfmt_last_error(self._h).decode())

    # ---- setup ----------------------------------------------------------------------------------
    def set_stream(self, stream_handle):
        self._chk(self._L.mpfmt_set_stream(self._h, C.c_void_p(stream_handle)))

    def set_state_bounds(self, ss_lo, ss_hi):
        """BoundedStateSpace lo / hi of any state dimension (after upload_shapes2d: the bounds of a steering space over the 2-D world)."""
        lo = np.ascontiguousarray(ss_lo, dtype=np.float64); hi = np.ascontiguousarray(ss_hi, dtype=np.float64)
        self._chk(self._L.mpfmt_set_state_bounds(self._h, _dp(lo), _dp(hi), len(lo)))

    def set_shard(self, rank, world):
        self._chk(self._L.mpfmt_set_shard(self._h, int(rank), int(world)))

    def upload_samples(self, X):
        """X: (N, d) float64 C-contiguous == Julia d x N column-major."""
        X = np.ascontiguousarray(X, dtype=np.float64)
        if X.ndim != 2:
            raise ValueError("X must be (N, d)")
        self._chk(self._L.mpfmt_upload_samples(self._h, _dp(X), X.shape[0], X.shape[1]))
        self.N, self.d = X.shape
        self.nnz = None

    def upload_samples_device(self, ptr, N, d):
        """The same from a device pointer (N x d float64, C-contiguous, on this ctx's GPU): no PCIe crossing."""
        self._chk(self._L.mpfmt_upload_samples_device(self._h, C.c_void_p(int(ptr)), int(N), int(d)))
        self.N, self.d = int(N), int(d)
        self.nnz = None

    def upload_boxes(self, lohi, ss_lo=None, ss_hi=None, dw=None):
        """lohi: (M, 2, dw) float64 (lo then hi per box)."""
        lohi = np.ascontiguousarray(lohi, dtype=np.float64)
        if lohi.size == 0:
            if dw is None:
                dw = lohi.shape[2] if (lohi.ndim == 3 and lohi.shape[2] > 0) else (len(ss_lo) if ss_lo is not None else self.d)
            lohi = np.zeros((0, 2, dw))
        if lohi.ndim != 3 or lohi.shape[1] != 2:
            raise ValueError("lohi must be (M, 2, dw)")
        M, _, dw = lohi.shape
        lo = None if ss_lo is None else np.ascontiguousarray(ss_lo, dtype=np.float64)
        hi = None if ss_hi is None else np.ascontiguousarray(ss_hi, dtype=np.float64)
        ds = 0 if lo is None else lo.size
        self._chk(self._L.mpfmt_upload_boxes(self._h, _dp(lohi), M, dw, _dp(lo), _dp(hi), ds))
        self.dw = dw

    def upload_shapes2d(self, shapes, ss_lo=None, ss_hi=None):
        """2-D SAT world: shapes = [("circle", (cx, cy), r) | ("polygon", [(x, y), ...]), ...] (a flat Compound2D)."""
        kinds = np.array([0 if s[0] == "circle" else 1 for s in shapes], dtype=np.int32)
        nv = np.array([0 if s[0] == "circle" else len(s[1]) for s in shapes], dtype=np.int32)
        parts = [np.array([s[1][0], s[1][1], s[2]], dtype=np.float64) if s[0] == "circle"
                 else np.ascontiguousarray(s[1], dtype=np.float64).reshape(-1) for s in shapes]
        data = np.concatenate(parts) if parts else np.zeros(1)
        lo = None if ss_lo is None else np.ascontiguousarray(ss_lo, dtype=np.float64)
        hi = None if ss_hi is None else np.ascontiguousarray(ss_hi, dtype=np.float64)
        i32 = C.POINTER(C.c_int32)
        self._chk(self._L.mpfmt_upload_shapes2d(self._h, len(shapes), kinds.ctypes.data_as(i32), nv.ctypes.data_as(i32), _dp(data),
                                                None if lo is None else _dp(lo), None if hi is None else _dp(hi)))
        self.dw = 2

    # ---- r-disc ---------------------------------------------------------------------------------
    def rdisc_count(self, r):
        colptr = np.empty(self.N + 1, dtype=np.int64)
        nnz = C.c_int64()
        self._chk(self._L.mpfmt_rdisc_count(self._h, float(r), _ip(colptr), C.byref(nnz)))
        self.nnz = nnz.value
        return colptr, nnz.value

    def rdisc_fill(self):
        nnz = self.nnz
        rowval = np.empty(max(nnz, 1), dtype=np.int64)
        nzval = np.empty(max(nnz, 1), dtype=np.float64)
        self._chk(self._L.mpfmt_rdisc_fill(self._h, _ip(rowval), _dp(nzval)))
        return rowval[:nnz], nzval[:nnz]

    def rdisc_graph(self, r):
        """(colptr, rowval, nzval), all 1-based like SparseMatrixCSC."""
        colptr, _ = self.rdisc_count(r)
        rowval, nzval = self.rdisc_fill()
        return colptr, rowval, nzval

    def rdisc_query(self, v, r, cap=None):
        cap = self.N if cap is None else cap
        inds = np.empty(max(cap, 1), dtype=np.int64)
        ds = np.empty(max(cap, 1), dtype=np.float64)
        k = C.c_int64()
        self._chk(self._L.mpfmt_rdisc_query(self._h, int(v), float(r), _ip(inds), _dp(ds), cap, C.byref(k)))
        return inds[:k.value].copy(), ds[:k.value].copy()

    # ---- sweeps ---------------------------------------------------------------------------------
    def points_free(self, idx=None):
        n = self.N if idx is None else len(idx)
        idx = None if idx is None else np.ascontiguousarray(idx, dtype=np.int64)
        mask = np.zeros(max(nwords(n), 1), dtype=np.uint64)
        self._chk(self._L.mpfmt_points_free(self._h, _ip(idx), n, _up(mask)))
        return mask[:nwords(n)]

    def edges_free(self, src, dst):
        src = np.ascontiguousarray(src, dtype=np.int64)
        dst = np.ascontiguousarray(dst, dtype=np.int64)
        E = src.size
        mask = np.zeros(max(nwords(E), 1), dtype=np.uint64)
        self._chk(self._L.mpfmt_edges_free(self._h, _ip(src), _ip(dst), E, _up(mask)))
        return mask[:nwords(E)]

    def graph_edges_free(self):
        mask = np.zeros(max(nwords(self.nnz), 1), dtype=np.uint64)
        self._chk(self._L.mpfmt_graph_edges_free(self._h, _up(mask)))
        return mask[:nwords(self.nnz)]

    def states_free(self, P):
        P = np.ascontiguousarray(P, dtype=np.float64)
        n = P.shape[0]
        mask = np.zeros(max(nwords(n), 1), dtype=np.uint64)
        self._chk(self._L.mpfmt_states_free(self._h, _dp(P), n, _up(mask)))
        return mask[:nwords(n)]

    def motions_free(self, P, Q):
        P = np.ascontiguousarray(P, dtype=np.float64)
        Q = np.ascontiguousarray(Q, dtype=np.float64)
        n = P.shape[0]
        mask = np.zeros(max(nwords(n), 1), dtype=np.uint64)
        self._chk(self._L.mpfmt_motions_free(self._h, _dp(P), _dp(Q), n, _up(mask)))
        return mask[:nwords(n)]

    # ---- Euclidean steer (geometric.jl:18-19) -----------------------------------------------------
    def euclid_steer(self, src, dst):
        """steering_control per edge: (t = |w - v|, u = unit direction) for 1-based src -> dst."""
        src = np.ascontiguousarray(src, dtype=np.int64); dst = np.ascontiguousarray(dst, dtype=np.int64)
        E = src.size
        t = np.empty(max(E, 1)); u = np.empty((max(E, 1), self.d))
        self._chk(self._L.mpfmt_euclid_steer(self._h, _ip(src), _ip(dst), E, _dp(t), _dp(u)))
        return t[:E], u[:E]

    def euclid_propagate(self, src, t, u, s=None):
        """propagate(M, V[src], StepControl(t, u)[, s])."""
        src = np.ascontiguousarray(src, dtype=np.int64)
        t = np.ascontiguousarray(t, dtype=np.float64); u = np.ascontiguousarray(u, dtype=np.float64)
        s = None if s is None else np.ascontiguousarray(s, dtype=np.float64)
        E = src.size
        out = np.empty((max(E, 1), self.d))
        self._chk(self._L.mpfmt_euclid_propagate(self._h, _ip(src), E, _dp(t), _dp(u), _dp(s), _dp(out)))
        return out[:E]

    # ---- expand / solve ---------------------------------------------------------------------------
    def expand(self, W, H, F, Cc, zs, cap=None):
        W = np.ascontiguousarray(W, dtype=np.uint64)
        H = np.ascontiguousarray(H, dtype=np.uint64)
        F = None if F is None else np.ascontiguousarray(F, dtype=np.uint64)
        Cc = np.ascontiguousarray(Cc, dtype=np.float64)
        zs = np.ascontiguousarray(zs, dtype=np.int64)
        cap = self.N if cap is None else cap
        xs = np.empty(max(cap, 1), dtype=np.int64)
        ym = np.empty(max(cap, 1), dtype=np.int64)
        cm = np.empty(max(cap, 1), dtype=np.float64)
        fr = np.empty(max(cap, 1), dtype=np.uint8)
        nx = C.c_int64()
        self._chk(self._L.mpfmt_expand(self._h, _up(W), _up(H), _up(F), _dp(Cc), _ip(zs), zs.size, _ip(xs), _ip(ym),
                                       _dp(cm), fr.ctypes.data_as(c_u8_p), cap, C.byref(nx)))
        n = nx.value
        return xs[:n].copy(), ym[:n].copy(), cm[:n].copy(), fr[:n].astype(bool)

    def fmtstar(self, r, goal_kind, goal_params, init_idx=1, checkpts=True):
        g = np.ascontiguousarray(goal_params, dtype=np.float64)
        A = np.empty(max(self.N, 1), dtype=np.int64)
        Cc = np.empty(max(self.N, 1), dtype=np.float64)
        path = np.empty(max(self.N, 1), dtype=np.int64)
        res = FmtResult()
        self._chk(self._L.mpfmt_fmtstar(self._h, float(r), int(init_idx), int(bool(checkpts)), int(goal_kind), _dp(g),
                                        _ip(A), _dp(Cc), _ip(path), C.byref(res)))
        return dict(status=int(res.status), cost=float(res.cost), z=int(res.z),
                    collision_checks=int(res.collision_checks), nnz=int(res.nnz),
                    ms_graph=res.ms_graph, ms_sweep=res.ms_sweep, ms_host_loop=res.ms_host_loop,
                    A=A[:self.N], C=Cc[:self.N], path=path[:res.path_len].copy())

    # ---- wavefront solve (recursion on the device) -------------------------------------------------
    @staticmethod
    def _wf_info(i):
        return {k: getattr(i, k) for k, _ in WfInfo._fields_}

    def _fmt_out(self, res, A, Cc, path):
        return dict(status=int(res.status), cost=float(res.cost), z=int(res.z), collision_checks=int(res.collision_checks),
                    nnz=int(res.nnz), ms_graph=res.ms_graph, ms_sweep=res.ms_sweep, ms_host_loop=res.ms_host_loop,
                    A=None if A is None else A[:self.N], C=None if Cc is None else Cc[:self.N], path=path[:res.path_len].copy())

    def fmtstar_wavefront(self, r, goal_kind, goal_params, band=0.0, single=False, eager=False, lazy=False, init_idx=1, checkpts=True,
                          want_tree=True):
        """fmtstar! with the recursion on the device (include/mpfmt.h): band = cost width of a batch; single = one node per
        step (the reference's order exactly)."""
        g = np.ascontiguousarray(goal_params, dtype=np.float64)
        A = np.empty(max(self.N, 1), dtype=np.int64) if want_tree else None
        Cc = np.empty(max(self.N, 1), dtype=np.float64) if want_tree else None
        path = np.empty(max(self.N, 1), dtype=np.int64)
        res, info = FmtResult(), WfInfo()
        flags = (WF_SINGLE if single else 0) | (WF_EAGER if eager else 0) | (WF_LAZY if lazy else 0)
        self._chk(self._L.mpfmt_fmtstar_wavefront(self._h, float(r), int(init_idx), int(bool(checkpts)), int(goal_kind), _dp(g), float(band),
                                                  flags, _ip(A), _dp(Cc), _ip(path), C.byref(res), C.byref(info)))
        out = self._fmt_out(res, A, Cc, path)
        out["info"] = self._wf_info(info)
        return out

    def wf_begin(self, r, goal_kind, goal_params, band=0.0, single=False, eager=False, init_idx=1, checkpts=True, lazy=False):
        g = np.ascontiguousarray(goal_params, dtype=np.float64)
        flags = (WF_SINGLE if single else 0) | (WF_EAGER if eager else 0) | (WF_LAZY if lazy else 0)
        self._chk(self._L.mpfmt_wf_begin(self._h, float(r), int(init_idx), int(bool(checkpts)), int(goal_kind), _dp(g), float(band), flags))

    def wf_step(self):
        info = WfInfo()
        self._chk(self._L.mpfmt_wf_step(self._h, C.byref(info)))
        return self._wf_info(info)

    def wf_state(self):
        """(W, H, C, A): packed uint64 masks, costs, parents (1-based, 0 = none)."""
        nw = max(nwords(self.N), 1)
        W = np.zeros(nw, dtype=np.uint64); H = np.zeros(nw, dtype=np.uint64)
        Cc = np.empty(max(self.N, 1)); A = np.empty(max(self.N, 1), dtype=np.int64)
        self._chk(self._L.mpfmt_wf_state(self._h, _up(W), _up(H), _dp(Cc), _ip(A)))
        return W[:nwords(self.N)], H[:nwords(self.N)], Cc[:self.N], A[:self.N]

    def wf_batch(self):
        zs = np.empty(max(self.N, 1), dtype=np.int64)
        n = C.c_int64()
        self._chk(self._L.mpfmt_wf_batch(self._h, _ip(zs), self.N, C.byref(n)))
        return zs[:n.value].copy()

    def wf_triples(self):
        x = np.empty(max(self.N, 1), dtype=np.int64); y = np.empty(max(self.N, 1), dtype=np.int64); c = np.empty(max(self.N, 1))
        n = C.c_int64()
        self._chk(self._L.mpfmt_wf_triples(self._h, self.N, _ip(x), _ip(y), _dp(c), C.byref(n)))
        return x[:n.value].copy(), y[:n.value].copy(), c[:n.value].copy()

    def wf_commit(self, x, y, c):
        x = np.ascontiguousarray(x, dtype=np.int64); y = np.ascontiguousarray(y, dtype=np.int64); c = np.ascontiguousarray(c, dtype=np.float64)
        self._chk(self._L.mpfmt_wf_commit(self._h, len(x), _ip(x), _ip(y), _dp(c)))

    def wf_finish(self, want_tree=True):
        A = np.empty(max(self.N, 1), dtype=np.int64) if want_tree else None
        Cc = np.empty(max(self.N, 1), dtype=np.float64) if want_tree else None
        path = np.empty(max(self.N, 1), dtype=np.int64)
        res = FmtResult()
        self._chk(self._L.mpfmt_wf_finish(self._h, _ip(A), _dp(Cc), _ip(path), C.byref(res)))
        return self._fmt_out(res, A, Cc, path)

    # ---- multi-GPU exchange (RCCL behind the ABI) ---------------------------------------------------
    def comm_create(self, rank, world, uid):
        """uid: 128 bytes from comm_unique_id() of rank 0, handed to every rank by the host."""
        u = np.frombuffer(bytes(uid), dtype=np.uint8).copy()
        assert u.size == COMM_ID_BYTES
        self._chk(self._L.mpfmt_comm_create(self._h, int(rank), int(world), u.ctypes.data_as(c_u8_p)))

    def comm_destroy(self):
        self._chk(self._L.mpfmt_comm_destroy(self._h))

    def allgather_free_mask_launch(self, cap_hint=0):
        self._chk(self._L.mpfmt_allgather_free_mask_launch(self._h, int(cap_hint)))

    def allgather_free_mask_finish(self, world, allow_retry=False):
        """(device ptr, stride in words, words per rank, nnz per rank); with allow_retry (single-thread drivers of several ctxs)
        None when the library asks for a relaunch of every ctx (MPFMT_RETRY)."""
        ptr, stride = C.c_void_p(), C.c_int64()
        words = np.zeros(world, dtype=np.int64); nnz = np.zeros(world, dtype=np.int64)
        rc = self._L.mpfmt_allgather_free_mask_finish(self._h, C.byref(ptr), C.byref(stride), _ip(words), _ip(nnz))
        if rc == RETRY and allow_retry:
            return None
        self._chk(rc)
        return ptr.value, stride.value, words, nnz

    def allgather_free_mask_relaunch(self):
        self._chk(self._L.mpfmt_allgather_free_mask_relaunch(self._h))

    def allgather_free_mask(self, world):
        """One RCCL all-gather of the per-shard free-edge masks: (device ptr, stride in words, words per rank, nnz per rank)."""
        self.allgather_free_mask_launch(0)
        return self.allgather_free_mask_finish(world)

    def mc_edges_collision(self, src, dst, sigma, rollouts, seed=0):
        """Colliding rollouts per edge (1-based src / dst): Monte-Carlo collision probability = hits / rollouts."""
        src = np.ascontiguousarray(src, dtype=np.int64); dst = np.ascontiguousarray(dst, dtype=np.int64)
        hits = np.zeros(max(len(src), 1), dtype=np.int64)
        self._chk(self._L.mpfmt_mc_edges_collision(self._h, _ip(src), _ip(dst), len(src), float(sigma), int(rollouts), int(seed), _ip(hits)))
        return hits[:len(src)]

    def mc_edges_collision_is(self, src, dst, sigma, rollouts, seed=0):
        """Importance-sampling estimate of the collision probability per edge (1-based src / dst): (probabilities, raw uint64 sums of
        the colliding rollouts' weights at 2^-40 -- what the scalar loop reproduces exactly)."""
        src = np.ascontiguousarray(src, dtype=np.int64); dst = np.ascontiguousarray(dst, dtype=np.int64)
        wsum = np.zeros(max(len(src), 1), dtype=np.uint64)
        self._chk(self._L.mpfmt_mc_edges_collision_is(self._h, _ip(src), _ip(dst), len(src), float(sigma), int(rollouts), int(seed),
                                                      wsum.ctypes.data_as(C.POINTER(C.c_uint64))))
        wsum = wsum[:len(src)]
        return wsum.astype(np.float64) / (2.0 ** 40) / max(int(rollouts), 1), wsum

    def mc_edges_collision_ais(self, src, dst, sigma, rollouts, seed=0):
        """ADAPTIVE importance sampling (pilot -> cross-entropy mean shift -> mixture): (probabilities, raw uint64 weight sums at 2^-40,
        shifts (E, 2 d) in noise units)."""
        src = np.ascontiguousarray(src, dtype=np.int64); dst = np.ascontiguousarray(dst, dtype=np.int64)
        wsum = np.zeros(max(len(src), 1), dtype=np.uint64)
        sh = np.zeros((max(len(src), 1), 2 * self.d))
        self._chk(self._L.mpfmt_mc_edges_collision_ais(self._h, _ip(src), _ip(dst), len(src), float(sigma), int(rollouts), int(seed),
                                                       wsum.ctypes.data_as(C.POINTER(C.c_uint64)), _dp(sh)))
        wsum = wsum[:len(src)]
        return wsum.astype(np.float64) / (2.0 ** 40) / max(int(rollouts), 1), wsum, sh[:len(src)]

    # ---- Dubins and Reeds-Shepp cars --------------------------------------------------------------
    def _car_graph(self, car, turn_radius, speed, r):
        colptr = np.empty(self.N + 1, dtype=np.int64)
        nnz = C.c_int64()
        self._chk(getattr(self._L, f"mpfmt_{car}_graph_count")(self._h, float(turn_radius), float(speed), float(r), _ip(colptr), C.byref(nnz)))
        self.nnz = n = nnz.value
        rowval = np.empty(max(n, 1), dtype=np.int64)
        nzval = np.empty(max(n, 1), dtype=np.float64)
        self._chk(getattr(self._L, f"mpfmt_{car}_graph_fill")(self._h, _ip(rowval), _dp(nzval)))
        return colptr, rowval[:n], nzval[:n]

    def _car_graph_edges_free(self, car):
        n = self.nnz
        mask = np.zeros(max(nwords(n), 1), dtype=np.uint64)
        nseg = np.zeros(max(n, 1), dtype=np.uint8)
        self._chk(getattr(self._L, f"mpfmt_{car}_graph_edges_free")(self._h, _up(mask), nseg.ctypes.data_as(c_u8_p)))
        return mask[:nwords(n)], nseg[:n]

    def _car_fmtstar(self, car, turn_radius, speed, r, goal_kind, goal_params, init_idx, checkpts):
        g = np.ascontiguousarray(goal_params, dtype=np.float64)
        A = np.empty(max(self.N, 1), dtype=np.int64)
        Cc = np.empty(max(self.N, 1), dtype=np.float64)
        path = np.empty(max(self.N, 1), dtype=np.int64)
        res = FmtResult()
        self._chk(getattr(self._L, f"mpfmt_{car}_fmtstar")(self._h, float(turn_radius), float(speed), float(r), int(init_idx),
                                                           int(bool(checkpts)), int(goal_kind), _dp(g), _ip(A), _dp(Cc), _ip(path),
                                                           C.byref(res)))
        return dict(status=int(res.status), cost=float(res.cost), z=int(res.z), collision_checks=int(res.collision_checks),
                    nnz=int(res.nnz), ms_graph=res.ms_graph, ms_sweep=res.ms_sweep, ms_host_loop=res.ms_host_loop,
                    A=A[:self.N], C=Cc[:self.N], path=path[:res.path_len].copy())

    def dubins_graph(self, turn_radius, speed, r):
        """Chopped backward sets of the Dubins quasi-metric, CSC 1-based: (colptr, rowval, nzval)."""
        return self._car_graph("dubins", turn_radius, speed, r)

    def dubins_graph_edges_free(self):
        return self._car_graph_edges_free("dubins")

    def dubins_steer(self, X0, X1, turn_radius, speed=1.0):
        X0 = np.ascontiguousarray(X0, dtype=np.float64); X1 = np.ascontiguousarray(X1, dtype=np.float64)
        n = len(X0)
        cost = np.empty(max(n, 1)); ctrl = np.empty((max(n, 1), 3, 3))
        self._chk(self._L.mpfmt_dubins_steer(self._h, _dp(X0), _dp(X1), n, float(turn_radius), float(speed), _dp(cost), _dp(ctrl)))
        return cost[:n], ctrl[:n]

    def dubins_fmtstar(self, turn_radius, speed, r, goal_kind, goal_params, init_idx=1, checkpts=True):
        return self._car_fmtstar("dubins", turn_radius, speed, r, goal_kind, goal_params, init_idx, checkpts)

    def reedsshepp_graph(self, turn_radius, speed, r):
        """Chopped Reeds-Shepp neighbourhoods, CSC 1-based: column v = rows w with nzval = reedsshepp(v, w)."""
        return self._car_graph("reedsshepp", turn_radius, speed, r)

    def reedsshepp_graph_edges_free(self):
        return self._car_graph_edges_free("reedsshepp")

    def reedsshepp_steer(self, X0, X1, turn_radius, speed=1.0):
        """(cost, controls (n,5,3), nsegs)."""
        X0 = np.ascontiguousarray(X0, dtype=np.float64); X1 = np.ascontiguousarray(X1, dtype=np.float64)
        n = len(X0)
        cost = np.empty(max(n, 1)); ctrl = np.empty((max(n, 1), 5, 3)); nsegs = np.zeros(max(n, 1), dtype=np.int32)
        self._chk(self._L.mpfmt_reedsshepp_steer(self._h, _dp(X0), _dp(X1), n, float(turn_radius), float(speed), _dp(cost), _dp(ctrl),
                                                 nsegs.ctypes.data_as(C.POINTER(C.c_int32))))
        return cost[:n], ctrl[:n], nsegs[:n]

    def reedsshepp_fmtstar(self, turn_radius, speed, r, goal_kind, goal_params, init_idx=1, checkpts=True):
        return self._car_fmtstar("reedsshepp", turn_radius, speed, r, goal_kind, goal_params, init_idx, checkpts)

    # ---- closest obstacle points ----------------------------------------------------------------
    def closest(self, P, W=None):
        """closest(p, CC, W) per row of P: (d2min, vmin, kmin 1-based / 0, failures)."""
        P = np.ascontiguousarray(np.atleast_2d(P), dtype=np.float64)
        n, d = P.shape
        Wp = None if W is None else _dp(np.ascontiguousarray(W, dtype=np.float64).reshape(d, d))
        d2 = np.empty(max(n, 1)); v = np.empty((max(n, 1), d)); k = np.empty(max(n, 1), dtype=np.int64)
        fails = C.c_int64()
        self._chk(self._L.mpfmt_closest(self._h, _dp(P), n, Wp, _dp(d2), _dp(v), _ip(k), C.byref(fails)))
        return d2[:n], v[:n], k[:n], fails.value

    def closeR(self, P, W, r2):
        """closeR(p, CC, W, r2) per row of P: (ptr 1-based, obstacle 1-based, d2, v, failures)."""
        P = np.ascontiguousarray(np.atleast_2d(P), dtype=np.float64)
        n, d = P.shape
        Wa = np.ascontiguousarray(W, dtype=np.float64).reshape(d, d)
        ptr = np.empty(n + 1, dtype=np.int64)
        total = C.c_int64(); fails = C.c_int64()
        cap = max(4 * n, 1024)
        while True:
            idx = np.empty(cap, dtype=np.int64); d2 = np.empty(cap); v = np.empty((cap, d))
            rc = self._L.mpfmt_closeR(self._h, _dp(P), n, _dp(Wa), float(r2), _ip(ptr), cap, _ip(idx), _dp(d2), _dp(v),
                                      C.byref(total), C.byref(fails))
            if rc == ERR_CAPACITY:
                cap = total.value
                continue
            self._chk(rc)
            t = total.value
            return ptr, idx[:t], d2[:t], v[:t], fails.value

    def graph_import(self, r, colptr, rowval, nzval):
        """Install an exported graph (1-based CSC as returned by rdisc_graph) for the uploaded samples."""
        colptr = np.ascontiguousarray(colptr, dtype=np.int64)
        rowval = np.ascontiguousarray(rowval, dtype=np.int64)
        nzval = np.ascontiguousarray(nzval, dtype=np.float64)
        self._chk(self._L.mpfmt_graph_import(self._h, float(r), _ip(colptr), _ip(rowval), _dp(nzval)))
        self.nnz = int(colptr[-1] - 1)

    def path_free(self, P):
        """is_free_path(p, CC, SS) for the states P (n, d): (free, per-segment bits)."""
        P = np.ascontiguousarray(P, dtype=np.float64)
        n = P.shape[0]
        fr = C.c_int32()
        mask = np.zeros(max(nwords(max(n - 1, 0)), 1), dtype=np.uint64)
        self._chk(self._L.mpfmt_path_free(self._h, _dp(P), n, C.byref(fr), _up(mask)))
        return bool(fr.value), unpack_bits(mask, max(n - 1, 0))

    def sample_free(self, seed, N, init=None, goal_kind=0, goal_params=None, goal_ct=0, goal_bias=0.0):
        """sample_free!(P, N): N free samples drawn on the device (counter-based stream, sequential semantics), left
        uploaded in the context.  Returns (X, attempts)."""
        d = self.dw
        X = np.empty((max(int(N), 1), d), dtype=np.float64)
        att = C.c_int64()
        ini = None if init is None else np.ascontiguousarray(init, dtype=np.float64)
        g = None if goal_params is None else np.ascontiguousarray(goal_params, dtype=np.float64)
        self._chk(self._L.mpfmt_sample_free_biased(self._h, int(seed), int(N), None if ini is None else _dp(ini), int(goal_kind),
                                                   None if g is None else _dp(g), int(goal_ct), float(goal_bias), _dp(X), C.byref(att)))
        self.N, self.d = int(N), d
        return X[:N], int(att.value)

    # ---- double integrator ------------------------------------------------------------------------
    def di_graph(self, rho, r):
        """Sparse cost matrix of the double-integrator space, CSC 1-based: (colptr, rowval, nzval, tval)."""
        colptr = np.empty(self.N + 1, dtype=np.int64)
        nnz = C.c_int64()
        self._chk(self._L.mpfmt_di_graph_count(self._h, float(rho), float(r), _ip(colptr), C.byref(nnz)))
        self.nnz = n = nnz.value
        rowval = np.empty(max(n, 1), dtype=np.int64)
        nzval = np.empty(max(n, 1), dtype=np.float64)
        tval = np.empty(max(n, 1), dtype=np.float64)
        self._chk(self._L.mpfmt_di_graph_fill(self._h, _ip(rowval), _dp(nzval), _dp(tval)))
        return colptr, rowval[:n], nzval[:n], tval[:n]

    def di_graph_edges_free(self):
        n = self.nnz
        mask = np.zeros(max(nwords(n), 1), dtype=np.uint64)
        nseg = np.zeros(max(n, 1), dtype=np.uint8)
        self._chk(self._L.mpfmt_di_graph_edges_free(self._h, _up(mask), nseg.ctypes.data_as(c_u8_p)))
        return mask[:nwords(n)], nseg[:n]

    def di_graph_step_device(self, rho, r):
        """Double-integrator graph + 5-waypoint edge bits, outputs left in HBM (see include/mpfmt.h).  Returns nnz."""
        nnz = C.c_int64()
        self._chk(self._L.mpfmt_di_graph_step_device(self._h, float(rho), float(r), C.byref(nnz)))
        self.nnz = nnz.value
        return nnz.value

    def di_graph_device_ptrs(self):
        """(colptr, rowval, nzval, tval, free_mask, nseg) device addresses of the resident double-integrator graph."""
        p = [C.c_void_p() for _ in range(6)]
        self._chk(self._L.mpfmt_di_graph_device_ptrs(self._h, *[C.byref(x) for x in p]))
        return tuple(x.value for x in p)

    def di_steer(self, X0, X1, rho, r):
        X0 = np.ascontiguousarray(X0, dtype=np.float64)
        X1 = np.ascontiguousarray(X1, dtype=np.float64)
        n, ns = X0.shape
        cost = np.empty(max(n, 1)); t = np.empty(max(n, 1))
        self._chk(self._L.mpfmt_di_steer(self._h, _dp(X0), _dp(X1), n, ns // 2, float(rho), float(r), _dp(cost), _dp(t)))
        return cost[:n], t[:n]

    def di_fmtstar(self, rho, r, goal_kind, goal_params, init_idx=1, checkpts=True):
        g = np.ascontiguousarray(goal_params, dtype=np.float64)
        A = np.empty(max(self.N, 1), dtype=np.int64)
        Cc = np.empty(max(self.N, 1), dtype=np.float64)
        path = np.empty(max(self.N, 1), dtype=np.int64)
        res = FmtResult()
        self._chk(self._L.mpfmt_di_fmtstar(self._h, float(rho), float(r), int(init_idx), int(bool(checkpts)), int(goal_kind),
                                           _dp(g), _ip(A), _dp(Cc), _ip(path), C.byref(res)))
        return dict(status=int(res.status), cost=float(res.cost), z=int(res.z), collision_checks=int(res.collision_checks),
                    nnz=int(res.nnz), ms_graph=res.ms_graph, ms_sweep=res.ms_sweep, ms_host_loop=res.ms_host_loop,
                    A=A[:self.N], C=Cc[:self.N], path=path[:res.path_len].copy())

    def car_fmtstar_wavefront(self, car, turn_radius, speed, r, goal_kind, goal_params, band=0.0, single=False, init_idx=1, checkpts=True):
        """dubins / reedsshepp planner with the recursion on the device."""
        g = np.ascontiguousarray(goal_params, dtype=np.float64)
        A = np.empty(max(self.N, 1), dtype=np.int64); Cc = np.empty(max(self.N, 1), dtype=np.float64)
        path = np.empty(max(self.N, 1), dtype=np.int64)
        res, info = FmtResult(), WfInfo()
        self._chk(getattr(self._L, f"mpfmt_{car}_fmtstar_wavefront")(self._h, float(turn_radius), float(speed), float(r), int(init_idx),
                                                                     int(bool(checkpts)), int(goal_kind), _dp(g), float(band),
                                                                     WF_SINGLE if single else 0, _ip(A), _dp(Cc), _ip(path), C.byref(res),
                                                                     C.byref(info)))
        out = self._fmt_out(res, A, Cc, path)
        out["info"] = self._wf_info(info)
        return out

    def di_fmtstar_wavefront(self, rho, r, goal_kind, goal_params, band=0.0, single=False, init_idx=1, checkpts=True, want_tree=True):
        """di_fmtstar with the recursion on the device (directed wavefront form)."""
        g = np.ascontiguousarray(goal_params, dtype=np.float64)
        A = np.empty(max(self.N, 1), dtype=np.int64) if want_tree else None
        Cc = np.empty(max(self.N, 1), dtype=np.float64) if want_tree else None
        path = np.empty(max(self.N, 1), dtype=np.int64)
        res, info = FmtResult(), WfInfo()
        self._chk(self._L.mpfmt_di_fmtstar_wavefront(self._h, float(rho), float(r), int(init_idx), int(bool(checkpts)), int(goal_kind), _dp(g),
                                                     float(band), WF_SINGLE if single else 0, _ip(A), _dp(Cc), _ip(path), C.byref(res),
                                                     C.byref(info)))
        out = self._fmt_out(res, A, Cc, path)
        out["info"] = self._wf_info(info)
        return out

    # ---- device-resident -------------------------------------------------------------------------
    def graph_build_device(self, r):
        nnz = C.c_int64()
        self._chk(self._L.mpfmt_graph_build_device(self._h, float(r), C.byref(nnz)))
        self.nnz = nnz.value
        return nnz.value

    def graph_sweep_device(self):
        self._chk(self._L.mpfmt_graph_sweep_device(self._h))

    def graph_step_device(self, r):
        """graph_build_device + graph_sweep_device with one host synchronisation (see include/mpfmt.h)."""
        nnz = C.c_int64()
        self._chk(self._L.mpfmt_graph_step_device(self._h, float(r), C.byref(nnz)))
        self.nnz = nnz.value
        return nnz.value

    def graph_step_launch(self, r):
        self._chk(self._L.mpfmt_graph_step_launch(self._h, float(r)))

    def graph_step_finish(self):
        nnz = C.c_int64()
        self._chk(self._L.mpfmt_graph_step_finish(self._h, C.byref(nnz)))
        self.nnz = nnz.value
        return nnz.value

    def graph_device_ptrs(self):
        p = [C.c_void_p() for _ in range(4)]
        self._chk(self._L.mpfmt_graph_device_ptrs(self._h, *[C.byref(x) for x in p]))
        return tuple(x.value for x in p)

    def graph_export(self, pinned=True, want_mask=True):
        """The resident graph + mask (as a step left them) in the ABI's host format: (colptr, rowval, nzval, mask, GB/s).  pinned: the
        destinations come from mpfmt_pinned_alloc (the copies then run at link speed); the arrays returned are copies."""
        N = self.N
        nnz = self.stat("nnz")
        words = (nnz + 63) // 64
        sizes = [8 * (N + 1), 8 * max(nnz, 1), 8 * max(nnz, 1), 8 * max(words, 1)]
        rate = C.c_double()
        if pinned:
            ptrs = []
            for b in sizes:
                p = C.c_void_p()
                rc = self._L.mpfmt_pinned_alloc(b, C.byref(p))
                if rc != 0:
                    raise MPFMTError(rc, "mpfmt_pinned_alloc failed")
                ptrs.append(p)
            try:
                self._chk(self._L.mpfmt_graph_export(self._h, C.cast(ptrs[0], c_i64_p), C.cast(ptrs[1], c_i64_p), C.cast(ptrs[2], c_d_p),
                                                     C.cast(ptrs[3], c_u64_p) if want_mask else None, C.byref(rate)))
                colptr = np.ctypeslib.as_array(C.cast(ptrs[0], c_i64_p), shape=(N + 1,)).copy()
                rowval = np.ctypeslib.as_array(C.cast(ptrs[1], c_i64_p), shape=(max(nnz, 1),))[:nnz].copy()
                nzval = np.ctypeslib.as_array(C.cast(ptrs[2], c_d_p), shape=(max(nnz, 1),))[:nnz].copy()
                mask = np.ctypeslib.as_array(C.cast(ptrs[3], c_u64_p), shape=(max(words, 1),))[:words].copy() if want_mask else None
            finally:
                for p in ptrs:
                    self._L.mpfmt_pinned_free(p)
        else:
            colptr = np.empty(N + 1, dtype=np.int64); rowval = np.empty(max(nnz, 1), dtype=np.int64)
            nzval = np.empty(max(nnz, 1), dtype=np.float64); mask = np.zeros(max(words, 1), dtype=np.uint64)
            self._chk(self._L.mpfmt_graph_export(self._h, _ip(colptr), _ip(rowval), _dp(nzval), _up(mask) if want_mask else None, C.byref(rate)))
            rowval, nzval, mask = rowval[:nnz], nzval[:nnz], (mask[:words] if want_mask else None)
        return colptr, rowval, nzval, mask, rate.value

    def graph_export_arena(self, want_mask=True, copy=True):
        """mpfmt_graph_export_pinned: the resident graph + mask into the ctx's own page-locked arena (kept across calls: only the first
        export of a ctx pays for page-locking).  copy=False returns views of the arena, valid until the next export / close."""
        cp, rv, nz, mk = c_i64_p(), c_i64_p(), c_d_p(), c_u64_p()
        nnz = C.c_int64(); rate = C.c_double()
        self._chk(self._L.mpfmt_graph_export_pinned(self._h, C.byref(cp), C.byref(rv), C.byref(nz), C.byref(mk) if want_mask else None,
                                                    C.byref(nnz), C.byref(rate)))
        n = nnz.value; words = (n + 63) // 64
        colptr = np.ctypeslib.as_array(cp, shape=(self.N + 1,))
        rowval = np.ctypeslib.as_array(rv, shape=(max(n, 1),))[:n]
        nzval = np.ctypeslib.as_array(nz, shape=(max(n, 1),))[:n]
        mask = np.ctypeslib.as_array(mk, shape=(max(words, 1),))[:words] if want_mask else None
        if copy:
            colptr, rowval, nzval = colptr.copy(), rowval.copy(), nzval.copy()
            mask = mask.copy() if mask is not None else None
        return colptr, rowval, nzval, mask, rate.value

    def rdisc_stream(self, r, Cc=None, H=None, want_free=False):
        """Per-column reductions of the r-disc graph without storing it: dict(deg, nnz[, parent (1-based, 0 = none), cost][, free_deg])."""
        N = self.N
        deg = np.zeros(max(N, 1), dtype=np.int64); fdeg = np.zeros(max(N, 1), dtype=np.int64)
        par = np.zeros(max(N, 1), dtype=np.int64); cost = np.zeros(max(N, 1), dtype=np.float64)
        nnz = C.c_int64()
        Cc_ = None if Cc is None else np.ascontiguousarray(Cc, dtype=np.float64)
        H_ = None if H is None else np.ascontiguousarray(H, dtype=np.uint64)
        self._chk(self._L.mpfmt_rdisc_stream(self._h, float(r), _dp(Cc_), _up(H_), int(bool(want_free)), _ip(deg), _ip(fdeg), _ip(par), _dp(cost),
                                             C.byref(nnz)))
        out = {"deg": deg[:N], "nnz": nnz.value}
        if Cc is not None:
            out["parent"], out["cost"] = par[:N], cost[:N]
        if want_free:
            out["free_deg"] = fdeg[:N]
        return out

    def shard_info(self):
        a, b, n = C.c_int64(), C.c_int64(), C.c_int64()
        self._chk(self._L.mpfmt_shard_info(self._h, C.byref(a), C.byref(b), C.byref(n)))
        return a.value, b.value, n.value

    def timing_reset(self):
        self._chk(self._L.mpfmt_timing_reset(self._h))

    def timing(self, name):
        ms, n = C.c_double(), C.c_int64()
        self._chk(self._L.mpfmt_timing_get(self._h, name.encode(), C.byref(ms), C.byref(n)))
        return ms.value, n.value

    def set_option(self, name, value):
        self._chk(self._L.mpfmt_set_option(self._h, name.encode(), int(value)))

    def stat(self, name):
        v = C.c_int64()
        self._chk(self._L.mpfmt_get_stat(self._h, name.encode(), C.byref(v)))
        return v.value

    def graph_stats(self):
        v = [C.c_int64() for _ in range(4)]
        self._chk(self._L.mpfmt_graph_stats(self._h, *[C.byref(x) for x in v]))
        return dict(pairs_tested=v[0].value, tiles=v[1].value, slices=v[2].value, cells=v[3].value)
