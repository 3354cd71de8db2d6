"""CPU tests of the host-side pieces that need no GPU: synthetic workloads, radius rule, mirror types."""
import numpy as np

import motionplanning_jl_amd as mp


def test_workloads_are_deterministic_and_shaped_like_baseline():
    a, b = mp.workloads.cfg1(), mp.workloads.cfg1()
    assert np.array_equal(a.X, b.X) and np.array_equal(a.lohi, b.lohi)
    assert (a.N, a.d, a.M) == (1000, 2, 20)
    assert np.array_equal(a.X[0], [0.1, 0.1]) and np.array_equal(a.X[-1], [0.9, 0.9])
    # boxes never contain init or goal
    for p in (a.X[0], a.X[-1]):
        assert not np.any(np.all((a.lohi[:, 0] <= p) & (p <= a.lohi[:, 1]), axis=1))
    w = mp.workloads.cfg2(2000)
    assert (w.d, w.M) == (6, 200) and abs(w.r - mp.workloads.fmt_radius(1.0, 6, 1.0, 2000)) < 1e-15
    assert abs(mp.workloads.fmt_radius(1.0, 6, 1.0, 1_000_000) - 0.17479) < 1e-5
    c4 = mp.workloads.cfg4(500)
    assert c4.X.shape == (500, 4) and np.all(np.abs(c4.X[:, 2:]) <= 0.5)


def test_mirror_types_without_a_device():
    SS = mp.UnitHypercube(3)
    assert mp.dim(SS) == 3 and mp.volume(SS) == 1.0
    DI = mp.DoubleIntegrator(2, vmax=0.5)
    assert mp.dim(DI) == 4 and DI.workspace_dim == 2 and np.allclose(DI.lo, [0, 0, -0.5, -0.5])
    CC = mp.PointRobotNDBoxes([mp.BoxBounds([[0.4, 0.5], [0.19, 0.35]])])          # BoxBounds(lohi::Matrix)
    assert np.allclose(CC.boxes[0].lo, [0.4, 0.19]) and np.allclose(CC.boxes[0].hi, [0.5, 0.35])
    assert CC.lohi().shape == (1, 2, 2) and CC.count == 0
    g = mp.BallGoal([0.9, 0.9], 0.05)
    assert np.allclose(g.params(), [0.9, 0.9, 0.05]) and g.kind == mp._lib.GOAL_BALL
    rng = np.random.default_rng(0)
    s = mp.sample_space(SS, rng, 10)
    assert s.shape == (10, 3) and np.all((0 <= s) & (s <= 1))
    v = g.sample(rng)
    assert np.linalg.norm(v - g.center) <= g.radius


def test_bit_packing_matches_bitvector_chunks():
    bits = np.zeros(200, bool); bits[[0, 63, 64, 199]] = True
    m = mp._lib.pack_bits(bits)
    assert m[0] == (1 | (1 << 63)) and m[1] == 1 and m[3] == (1 << 7)
    assert np.array_equal(mp._lib.unpack_bits(m, 200), bits)


def _host_case(N, d, M, seed, goal_radius):
    from oracle import oracle as orc
    rng = np.random.default_rng(seed)
    w = mp.workloads.make("t", N, d, M, 0.05, 0.15, seed=seed, goal_radius=goal_radius)
    colptr, rowval, nzval = orc.rdisc_graph(w.X, w.r)                       # 0-based CSC, ascending rows
    efree = orc.graph_edges_free(w.X, colptr, rowval, w.lohi, w.ss_lo, w.ss_hi)
    F = orc.points_free(w.X, w.lohi, w.ss_lo, w.ss_hi)
    return orc, w, colptr, rowval, nzval, efree, F


def test_host_fmt_recursion_matches_oracle():
    """The C++ host recursion the library runs after the GPU phases (mpfmt_host_fmt_recursion, no device needed)
    against the oracle's recursion on the same graph: same tree, costs, path and collision count."""
    for (N, d, M, seed, gr) in [(1500, 2, 20, 11, 0.05), (3000, 3, 40, 12, 0.1), (2000, 6, 100, 13, 0.3)]:
        orc, w, colptr, rowval, nzval, efree, F = _host_case(N, d, M, seed, gr)
        goal = w.goal_params()
        for use_F in (True, False):
            want = orc.fmtstar_graph(w.X, colptr, rowval, nzval, efree, F if use_F else None, mp._lib.GOAL_BALL, goal,
                                     w.lohi, w.ss_lo, w.ss_hi, init_idx=0)
            got = mp._lib.host_fmt_recursion(w.X, colptr, rowval, nzval, efree, F if use_F else None, mp._lib.GOAL_BALL,
                                             goal, w.ss_lo, w.ss_hi, init_idx=1)
            assert got["status"] == want["status"] and got["z"] == want["z"] + 1
            assert got["cost"] == want["cost"] and got["collision_checks"] == want["collision_checks"]
            assert np.array_equal(got["A"], want["A"] + 1)                  # the oracle's parents are 0-based (-1 = none)
            assert np.array_equal(got["C"], want["C"])
            assert np.array_equal(got["path"], want["path"] + 1)


def test_host_fmt_recursion_rejects_bad_arguments():
    import pytest
    X = np.zeros((4, 2)); cp = np.zeros(5, np.int64); rv = np.zeros(1, np.int32); nz = np.zeros(1); m = np.zeros(1, np.uint64)
    with pytest.raises(mp.MPFMTError):
        mp._lib.host_fmt_recursion(X, cp, rv, nz, m, None, mp._lib.GOAL_POINT, [0.0, 0.0], init_idx=0)      # 1-based index
    with pytest.raises(mp.MPFMTError):
        mp._lib.host_fmt_recursion(X, cp, rv, nz, m, None, 7, [0.0, 0.0])                                     # unknown goal kind
    with pytest.raises(mp.MPFMTError):
        mp._lib.host_fmt_recursion(X, cp, rv, nz, m, None, mp._lib.GOAL_POINT, [0.0, 0.0], ss_lo=[0.0, 0.0])  # lo without hi


def test_bench_gpus_flag_is_honoured():
    """bench.py --gpus N must either start N ranks or fail loudly -- never run one rank and call it N.
    Without GPUs here: the parent refuses before spawning; a launcher whose WORLD_SIZE disagrees is refused too."""
    import os
    import subprocess
    import sys
    ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    bench = os.path.join(ROOT, "bench.py")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    p = subprocess.run([sys.executable, bench, "--gpus", "2", "--steps", "1", "--warmup", "0"], env=env, capture_output=True, text=True, timeout=300)
    assert p.returncode == 2 and "only 0 GPU" in p.stderr and p.stdout.strip() == ""
    env2 = dict(env, WORLD_SIZE="4", RANK="0", LOCAL_RANK="0")
    p = subprocess.run([sys.executable, bench, "--gpus", "2"], env=env2, capture_output=True, text=True, timeout=300)
    assert p.returncode == 2 and "does not match WORLD_SIZE" in p.stderr


def test_host_side_under_asan_ubsan(orc, tmp_path):
    """The host part of libmpfmt.so (mpfmt_host.cpp: sequential recursion, directed recursion, import validation) built with
    -fsanitize=address,undefined and run on an oracle-made graph: no sanitizer report, and the tree equals the oracle's."""
    import ctypes
    import os
    import subprocess
    ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    csrc = os.path.join(ROOT, "motionplanning.jl_amd", "csrc")
    exe = str(tmp_path / "host_asan")
    subprocess.check_call(["g++", "-O1", "-g", "-std=c++17", "-fsanitize=address,undefined", "-fno-sanitize-recover=all", "-ffp-contract=off",
                           os.path.join(ROOT, "tests", "asan", "host_asan.cpp"), os.path.join(csrc, "mpfmt_host.cpp"), "-o", exe])
    w = mp.workloads.make("t", 1500, 3, 30, 0.05, 0.12, seed=8, goal_radius=0.1)
    colptr, rowval, nzval = orc.rdisc_graph(w.X, w.r)
    emask = orc.graph_edges_free(w.X, colptr, rowval, w.lohi, w.ss_lo, w.ss_hi)
    F = orc.points_free(w.X, w.lohi, w.ss_lo, w.ss_hi)
    with open(tmp_path / "in.bin", "wb") as f:
        f.write(np.array([w.N, w.d, len(rowval), 1], dtype=np.int64).tobytes())
        f.write(w.X.tobytes()); f.write(colptr.astype(np.int64).tobytes()); f.write(rowval.astype(np.int32).tobytes())
        f.write(nzval.tobytes()); f.write(emask.tobytes()); f.write(F.tobytes())
        f.write(w.ss_lo.tobytes()); f.write(w.ss_hi.tobytes()); f.write(w.goal_params().tobytes())
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1")
    p = subprocess.run([exe, str(tmp_path / "in.bin"), str(tmp_path / "out.bin")], env=env, capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stdout + p.stderr
    assert "ERROR" not in p.stderr and "runtime error" not in p.stderr, p.stderr
    buf = open(tmp_path / "out.bin", "rb").read()
    R = mp._lib.FmtResult
    res = R.from_buffer_copy(buf, 0)
    o = ctypes.sizeof(R)
    N = w.N
    A = np.frombuffer(buf, np.int64, N, o); Cc = np.frombuffer(buf, np.float64, N, o + 8 * N)
    path = np.frombuffer(buf, np.int64, res.path_len, o + 16 * N)
    res2 = R.from_buffer_copy(buf, o + 16 * N + 8 * res.path_len)
    ref = orc.fmtstar_graph(w.X, colptr, rowval, nzval, emask, F, orc.GOAL_BALL, w.goal_params(), w.lohi, w.ss_lo, w.ss_hi)
    assert res.status == ref["status"] and res.collision_checks == ref["collision_checks"] and res.cost == ref["cost"]
    assert np.array_equal(A - 1, ref["A"]) and np.array_equal(Cc, ref["C"]) and np.array_equal(path - 1, ref["path"])
    # the directed recursion on a symmetric graph is the same recursion
    assert res2.status == res.status and res2.cost == res.cost and res2.collision_checks == res.collision_checks
