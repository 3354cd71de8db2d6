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
    env = dict(os.environ, MPFMT_RCCL_LIB=mock_lib, MPFMT_BENCH_ONE_DEVICE="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
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
    one = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1", "--workload", "cfg2", "--no-cpu-baseline", "--no-solve"],
                         capture_output=True, text=True, timeout=600)
    assert one.returncode == 0, one.stderr[-2000:]
    d1 = json.loads([l for l in one.stdout.splitlines() if l.startswith("{")][-1])
    assert d["config"]["nnz"] == d1["config"]["nnz"]                 # the shards partition the graph


def test_bench_starts_its_own_ranks(mock_lib):
    """`python bench.py --gpus 2` with no launcher around it: the parent spawns the ranks (before touching the GPU) and rank 0's
    JSON line comes back on its stdout."""
    env = dict(os.environ, MPFMT_RCCL_LIB=mock_lib, MPFMT_BENCH_ONE_DEVICE="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--workload", "cfg2",
                        "--no-cpu-baseline"], env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-3000:]
    d = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][-1])
    assert d["n_gpus"] == 2 and d["steps"] == 2 and d["value"] > 0
