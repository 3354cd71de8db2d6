"""motionplanning.jl_amd -- MI355X-native FMT* batch-expand hot path behind the MotionPlanning.jl plugin
surface (r-disc neighbour graph, segment-vs-AABB sweep, per-edge cost, FMT* expansion).

Layout: csrc/ (hand-written HIP for gfx950 + the C ABI of include/mpfmt.h -> libmpfmt.so),
_lib.py (ctypes binding), host mirror of the reference's Julia interface (statespaces, collision
checkers, near-neighbour sets, goals, problem, planner), workloads.py (synthetic BASELINE configs).
No CPU fallback exists for any compute entry point.
"""
from . import _lib, workloads, distributed, mirror  # noqa: F401
from .mirror import *  # noqa: F401,F403  (the reference's names: MPProblem, UnitHypercube, fmtstar_, ...)
from ._lib import Context, MPFMTError  # noqa: F401

__version__ = "0.1.0"
