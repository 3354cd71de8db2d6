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


def test_committed_counter_summary_belongs_to_this_build():
    """profiles/traffic.json (rocprofv3 --pmc, tools/pmc_traffic.py) is keyed to the sha256 of the library it was taken on; bench.py
    prints `roofline.traffic` only when that is the library it runs.  The committed summary must be the one of the library built from
    the committed sources (the build is reproducible: same compiler, same flags, same bytes)."""
    import hashlib
    import json
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    so = os.path.join(root, "motionplanning.jl_amd", "libmpfmt.so")
    tj = os.path.join(root, "profiles", "traffic.json")
    if not (os.path.exists(so) and os.path.exists(tj)):
        pytest.skip("no built library or no counter summary")
    sha = hashlib.sha256(open(so, "rb").read()).hexdigest()
    t = json.load(open(tj))
    assert t["lib_sha256"] == sha, "profiles/traffic.json was taken on another build: run tools/final_profiles.sh on the GPU box"
