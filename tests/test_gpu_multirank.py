"""GPU tests (-m gpu) of the MULTI-PROCESS exchange behind the C ABI on a one-GPU box: RCCL is replaced by a shared-memory
stand-in (tests/mock_rccl, selected with MPFMT_RCCL_LIB) because real RCCL refuses two ranks on one device.  What runs is the
library's own code: capacity agreement across ranks, launch / finish, the growth path, the per-wavefront triple exchange, and
bench.py's N > 1 path end to end."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


@pytest.fixture(scope="module")
def mock_lib(tmp_path_factory):
    so = str(tmp_path_factory.mktemp("mock") / "librccl_mock.so")
    subprocess.check_call(["g++", "-O1", "-shared", "-fPIC", os.path.join(ROOT, "tests", "mock_rccl", "mock_rccl.cpp"), "-I/opt/rocm/include",
                           "-D__HIP_PLATFORM_AMD__", "-L/opt/rocm/lib", "-lamdhip64", "-lrt", "-o", so])
    return so


def run_ranks(n, script, args, mock_lib, timeout=600):
    env = dict(os.environ, MPFMT_RCCL_LIB=mock_lib, MPFMT_ALLOW_RCCL_OVERRIDE="1", MPFMT_BENCH_ONE_DEVICE="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=%d" % n, "--master-addr", "127.0.0.1",
           "--master-port", str(free_port()), script] + args
    return subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=timeout)


@pytest.mark.parametrize("world", [2, 3])
def test_mask_gather_and_wavefront_exchange_across_processes(mock_lib, world):
    p = run_ranks(world, os.path.join(ROOT, "tests", "mp_multirank_worker.py"), [], mock_lib)
    assert p.returncode == 0, p.stdout[-3000:] + p.stderr[-3000:]
    assert "multirank ok: world %d" % world in p.stdout


def test_bench_two_ranks_through_the_c_abi(mock_lib):
    p = run_ranks(2, os.path.join(ROOT, "bench.py"), ["--gpus", "2", "--steps", "3", "--warmup", "2", "--workload", "cfg2", "--no-cpu-baseline"], mock_lib)
    assert p.returncode == 0, p.stdout[-3000:] + p.stderr[-3000:]
    line = [l for l in p.stdout.splitlines() if l.startswith("{")][-1]
    d = json.loads(line)
    assert d["n_gpus"] == 2 and d["config"]["parallelism"] == "shard2" and "C ABI" in d["config"]["exchange"]
    one = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "3", "--warmup", "2", "--workload", "cfg2", "--no-cpu-baseline", "--no-solve", "--no-cold"],
                         capture_output=True, text=True, timeout=600)
    assert one.returncode == 0, one.stderr[-2000:]
    d1 = json.loads([l for l in one.stdout.splitlines() if l.startswith("{")][-1])
    assert d["config"]["nnz"] == d1["config"]["nnz"]                 # the shards partition the graph


def test_bench_starts_its_own_ranks(mock_lib):
    """`python bench.py --gpus 2` with no launcher around it: the parent spawns the ranks (before touching the GPU) and rank 0's
    JSON line comes back on its stdout."""
    env = dict(os.environ, MPFMT_RCCL_LIB=mock_lib, MPFMT_ALLOW_RCCL_OVERRIDE="1", MPFMT_BENCH_ONE_DEVICE="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--workload", "cfg2",
                        "--no-cpu-baseline"], env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-3000:]
    d = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][-1])
    assert d["n_gpus"] == 2 and d["steps"] == 2 and d["value"] > 0


def run_group(world, mock_lib, args=(), timeout=900):
    env = dict(os.environ, MPFMT_RCCL_LIB=mock_lib, MPFMT_ALLOW_RCCL_OVERRIDE="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    return subprocess.run([sys.executable, os.path.join(ROOT, "tests", "sp_group_worker.py"), str(world)] + [str(a) for a in args],
                          env=env, capture_output=True, text=True, timeout=timeout)


@pytest.mark.parametrize("world", [2, 3])
def test_one_thread_drives_all_ranks(mock_lib, world):
    """The single-thread driver of include/mpfmt.h (one Julia process holding G handles): launches bracketed by
    mpfmt_group_begin / _end against a stand-in that defers grouped collectives like RCCL, the next step's kernels enqueued before
    the gather is finished, growth through MPFMT_RETRY / _relaunch."""
    p = run_group(world, mock_lib)
    assert p.returncode == 0, p.stdout[-3000:] + p.stderr[-3000:]
    assert "group ok: world %d" % world in p.stdout


def test_eight_ranks_at_north_star_size_on_one_gpu(mock_lib):
    """Dry run of the 8-GPU node: 8 shards of the N = 1e6 workload, all on this GPU, through the library's exchange: capacity
    agreement at 13 MB of masks, sum of the shards' nnz, every shard's mask against the unsharded one."""
    p = run_group(8, mock_lib, args=(1000000, 6), timeout=1500)
    assert p.returncode == 0, p.stdout[-3000:] + p.stderr[-3000:]
    assert "group ok: world 8, N 1000000" in p.stdout


def test_rccl_override_needs_the_second_opt_in(mock_lib):
    """MPFMT_RCCL_LIB alone must not substitute the collective library."""
    env = dict(os.environ, MPFMT_RCCL_LIB=mock_lib, HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.pop("MPFMT_ALLOW_RCCL_OVERRIDE", None)
    code = ("import sys; sys.path.insert(0, %r); import motionplanning_jl_amd as mp\n"
            "try:\n    mp._lib.comm_unique_id(); print('LOADED')\nexcept mp.MPFMTError as e:\n    print('REFUSED', e)\n" % ROOT)
    p = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
    assert "REFUSED" in p.stdout and "MPFMT_ALLOW_RCCL_OVERRIDE" in p.stdout, p.stdout + p.stderr
