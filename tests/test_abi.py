"""CPU tests of the boundary: libmpfmt.so loads, exports every symbol include/mpfmt.h declares, and fails
loudly (no CPU fallback) when no gfx950 device is present."""
import ctypes
import os
import re

import pytest

import motionplanning_jl_amd as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols():
    h = open(os.path.join(ROOT, "include", "mpfmt.h")).read()
    h = re.sub(r"/\*.*?\*/", "", h, flags=re.S)
    return sorted(set(re.findall(r"\b(mpfmt_[A-Za-z0-9_]+)\s*\(", h)))


def test_library_exports_every_declared_symbol():
    L = ctypes.CDLL(mp._lib.so_path())
    syms = header_symbols()
    assert len(syms) >= 24
    for s in syms:
        assert hasattr(L, s), "libmpfmt.so does not export %s" % s
    # and the Python binding table covers the header exactly
    assert sorted(n for n, _, _ in mp._lib.SYMBOLS) == syms


def test_library_exports_nothing_but_the_abi():
    """Built with -fvisibility=hidden + MPFMT_API: the dynamic symbol table holds the header's functions and nothing else (no C++
    helper, no kernel launch stub that another HIP library in the same process could interpose)."""
    import subprocess
    out = subprocess.check_output(["nm", "-D", "--defined-only", mp._lib.so_path()]).decode()
    defined = sorted(l.split()[-1] for l in out.splitlines() if len(l.split()) == 3 and l.split()[1] in "TtWwVvBbDdRr")
    funcs = sorted(l.split()[-1] for l in out.splitlines() if len(l.split()) == 3 and l.split()[1] == "T")
    assert funcs == header_symbols(), sorted(set(funcs) ^ set(header_symbols()))
    extra = [s for s in defined if s not in funcs]
    assert all(not s.startswith(("_Z", "mpfmt")) for s in extra), extra


def test_version_string():
    assert mp._lib.lib().mpfmt_version().decode().endswith("gfx950")


def test_no_cpu_fallback_without_device():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(mp.MPFMTError) as e:
        mp.Context(0)
    assert e.value.code == mp._lib.ERR_NODEVICE
    assert "no CPU fallback" in str(e.value)


def test_product_does_not_import_oracle():
    """The oracle is test infrastructure: nothing in the package may reference it."""
    pkg = os.path.join(ROOT, "motionplanning.jl_amd")
    for dp, _, fs in os.walk(pkg):
        for f in fs:
            if f.endswith((".py", ".hip", ".h", ".cpp", ".jl")) or f == "Makefile":
                txt = open(os.path.join(dp, f), errors="ignore").read()
                assert "liboracle" not in txt and "mpfmt_oracle" not in txt and "from oracle" not in txt \
                    and "import oracle" not in txt, os.path.join(dp, f)


def test_every_ccall_of_the_julia_glue_is_executed_by_a_c_caller():
    """julia/MPFmtHIP.jl cannot run here (no Julia): every symbol it ccalls must be called by tests/abi_c/abi_caller.c or abi_caller2.c,
    which the GPU suite builds with the ccall argument widths and runs."""
    jl = open(os.path.join(ROOT, "julia", "MPFmtHIP.jl")).read()
    used = set(re.findall(r"ccall\(\(:(mpfmt_[A-Za-z0-9_]+)", jl))
    csrc = open(os.path.join(ROOT, "tests", "abi_c", "abi_caller.c")).read() + open(os.path.join(ROOT, "tests", "abi_c", "abi_caller2.c")).read()
    called = set(re.findall(r"\b(mpfmt_[A-Za-z0-9_]+)\b", csrc))
    assert len(used) >= 30
    assert not (used - called), sorted(used - called)
    assert used <= set(header_symbols())


def test_committed_counter_summary_is_keyed_to_a_library():
    """profiles/traffic.json (rocprofv3 --pmc, tools/pmc_traffic.py) is keyed to the sha256 of the library it was taken on; bench.py
    prints `roofline.traffic` only when that is the library it runs (otherwise null).  Whether the key matches THIS build is checked where
    it matters -- by bench.py on the GPU box and by tools/final_profiles.sh, which regenerates the summary -- not by the CPU suite: a
    source or compiler change must not fail unrelated tests until someone re-profiles."""
    import json
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    tj = os.path.join(root, "profiles", "traffic.json")
    if not os.path.exists(tj):
        pytest.skip("no counter summary")
    t = json.load(open(tj))
    assert isinstance(t.get("lib_sha256"), str) and len(t["lib_sha256"]) == 64
    # ... but a stale summary is SAID: a warning in the CPU suite's report when the built library is not the profiled one
    import hashlib
    import warnings
    so = mp._lib.so_path()
    if os.path.exists(so):
        sha = hashlib.sha256(open(so, "rb").read()).hexdigest()
        if sha != t["lib_sha256"]:
            warnings.warn("profiles/traffic.json was taken on another build of libmpfmt.so (%s..., built: %s...): bench.py will print "
                          "roofline.traffic = null until tools/final_profiles.sh has been run on this build" % (t["lib_sha256"][:12], sha[:12]))
    for name, e in (json.load(open(os.path.join(root, "profiles", "valu_ops.json"))) if os.path.exists(os.path.join(root, "profiles", "valu_ops.json")) else {}).items():
        assert isinstance(e.get("lib_sha256"), str) and len(e["lib_sha256"]) == 64, name
