"""Synthetic workloads of SURVEY.md section 8(d): the configurations BASELINE.json names.

Samples i.i.d. uniform in [0,1]^d (fp64); V[1] = init; last sample = goal centre; obstacles = M AABBs,
centre ~ U[0,1]^d, half-width per axis ~ U[h_lo,h_hi], boxes containing init or goal rejected.
Random numbers come from SplitMix64 (Steele, Lea, Flood 2014; the generator xoshiro's authors seed with), used as a
counter-based stream: draw i (0-based) of seed s is mix(s + (i + 1) * 0x9E3779B97F4A7C15), i.e. exactly the i-th output of
the textbook generator, and u = (draw >> 11) * 2^-53.  Five lines in any language, so a Julia harness regenerates the bench
inputs bit for bit; tests/golden/stream_heads.json pins the first draws of every seed used here.
"""
import math
from dataclasses import dataclass

import numpy as np


@dataclass
class Workload:
    name: str
    X: np.ndarray          # (N, d)
    lohi: np.ndarray       # (M, 2, d)
    r: float
    init: np.ndarray
    goal_center: np.ndarray
    goal_radius: float
    ss_lo: np.ndarray
    ss_hi: np.ndarray
    seed: int = 0
    draw: object = None    # non-uniform sample sets: draw(stream, N, d) -> (N, d) samples (None: i.i.d. uniform)
    device_set: object = None      # sample sets drawn by the library's sampler: device_set(ctx, k) -> (N, d)
    goal_bias: float = 0.0

    @property
    def N(self):
        return self.X.shape[0]

    @property
    def d(self):
        return self.X.shape[1]

    @property
    def M(self):
        return self.lohi.shape[0]

    def goal_params(self):
        return np.concatenate([self.goal_center, [self.goal_radius]])


_GAMMA = np.uint64(0x9E3779B97F4A7C15)
_M1 = np.uint64(0xBF58476D1CE4E5B9)
_M2 = np.uint64(0x94D049BB133111EB)


def splitmix64(seed, n, offset=0):
    """Outputs offset .. offset+n-1 (0-based) of SplitMix64 seeded with `seed`, as uint64 (wrap-around arithmetic)."""
    with np.errstate(over="ignore"):
        i = np.arange(offset + 1, offset + n + 1, dtype=np.uint64)
        z = np.uint64(seed) + i * _GAMMA
        z = (z ^ (z >> np.uint64(30))) * _M1
        z = (z ^ (z >> np.uint64(27))) * _M2
        return z ^ (z >> np.uint64(31))


class Stream:
    """Sequential view of the counter-based stream: .random(shape) hands out the next prod(shape) uniforms in [0, 1), C order."""

    def __init__(self, seed):
        self.seed, self.pos = int(seed), 0

    def random(self, shape=None):
        n = int(np.prod(shape)) if shape is not None else 1
        u = (splitmix64(self.seed, n, self.pos) >> np.uint64(11)).astype(np.float64) * 2.0 ** -53
        self.pos += n
        return u.reshape(shape) if shape is not None else float(u[0])


def fmt_radius(rm, d, vol, N):
    """src/planners/fmt.jl:39, evaluated left to right like the Julia expression."""
    zeta = math.pi ** (d / 2) / math.gamma(d / 2 + 1)
    inner = 1 / d * vol / zeta * math.log(N) / N
    return rm * 2 * inner ** (1 / d)


def make_boxes(rng, M, d, h_lo, h_hi, keep_out):
    out = np.empty((M, 2, d))
    k = 0
    while k < M:
        c = rng.random(d)
        h = h_lo + (h_hi - h_lo) * rng.random(d)
        lo, hi = c - h, c + h
        if any(np.all((lo <= p) & (p <= hi)) for p in keep_out):
            continue
        out[k, 0], out[k, 1] = lo, hi
        k += 1
    return out


def make(name, N, d, M, h_lo, h_hi, seed, init_v=0.1, goal_v=0.9, goal_radius=0.15, rm=1.0, r=None):
    rng = Stream(seed)
    init = np.full(d, init_v)
    goal = np.full(d, goal_v)
    lohi = make_boxes(rng, M, d, h_lo, h_hi, [init, goal])
    X = rng.random((N, d))
    X[0] = init
    X[-1] = goal
    if r is None:
        r = fmt_radius(rm, d, 1.0, N)
    return Workload(name, X, lohi, float(r), init, goal, goal_radius, np.zeros(d), np.ones(d), seed)


def resample(w, k):
    """Sample set k of the same problem (k = 0: w.X itself): what a planner's k-th call on this world draws -- new i.i.d. samples of
    the same N, init first, goal centre last, same obstacles and radius.  Stream seed = w.seed + 1000 k (pinned like every other)."""
    if k == 0:
        return w.X
    rng = Stream(w.seed + 1000 * k)
    X = w.draw(rng, w.N, w.d) if w.draw else rng.random((w.N, w.d))
    X[0] = w.init
    X[-1] = w.goal_center
    return X


def cfg1(N=1000):
    """FMT* in 2-D, N=1000, 20 AABBs (BASELINE.json configs[0], the CPU-runnable plumbing case)."""
    return make("cfg1_2d_n1000_m20", N, 2, 20, 0.02, 0.08, seed=1, goal_radius=0.05)


def cfg2(N=100_000):
    """FMT* in R^6, N=100k, 200 AABBs (BASELINE.json configs[1])."""
    return make("cfg2_r6_n100k_m200", N, 6, 200, 0.10, 0.20, seed=2)


def north_star(N=1_000_000):
    """FMT* in R^6, N=1e6, 200 AABBs: the configuration BASELINE.json's metric is quoted on."""
    return make("ns_r6_n1m_m200", N, 6, 200, 0.10, 0.20, seed=3)


def cfg3(N=1_000_000, deg=256.0):
    """R^12 all-pairs r-disc graph (BASELINE.json configs[2]); r chosen for an expected interior degree."""
    d = 12
    zeta = math.pi ** (d / 2) / math.gamma(d / 2 + 1)
    r = (deg / (N * zeta)) ** (1.0 / d)
    return make("cfg3_r12_n1m_m200", N, d, 200, 0.20, 0.35, seed=4, r=r)


def _box_muller(rng, n):
    """n standard normals from 2 ceil(n/2) uniforms of the stream (Box-Muller, both branches)."""
    m = (n + 1) // 2
    u1 = 1.0 - rng.random(m)                       # (0, 1]
    u2 = rng.random(m)
    rad = np.sqrt(-2.0 * np.log(u1))
    return np.concatenate([rad * np.cos(2 * np.pi * u2), rad * np.sin(2 * np.pi * u2)])[:n]


def clustered_draw(frac=0.3, sigma=0.25, center=0.5):
    """Sample sets that are NOT uniform: a fraction `frac` of the samples from an isotropic Gaussian (sigma per axis, redrawn uniformly
    when it leaves the unit cube), the rest i.i.d. uniform -- at frac = 0.3, sigma = 0.25 in R^6 the density at the centre is ~5.7 x the
    uniform one and 0.7 x far from it: what a planner's sample set looks like after goal / obstacle-aware sampling.  The clustered samples
    are interleaved with the uniform ones (every sample decides by its own uniform), as a sampler with a bias probability emits them."""
    def draw(rng, N, d):
        X = rng.random((N, d))
        pick = rng.random(N) < frac
        G = center + sigma * _box_muller(rng, N * d).reshape(N, d)
        ok = pick & np.all((G >= 0.0) & (G < 1.0), axis=1)
        X[ok] = G[ok]
        return X
    return draw


def north_star_clustered(N=1_000_000):
    """The north star's world (R^6, 200 boxes, the fmt.jl:39 radius of N uniform samples) with CLUSTERED samples (clustered_draw): the
    throughput-under-non-uniform-density workload: ~1.6 x the edges of the uniform set, columns up to ~6 x the mean."""
    w = make("ns_r6_n1m_m200_clustered", N, 6, 200, 0.10, 0.20, seed=7)
    w.draw = clustered_draw()
    rng = Stream(w.seed + 500)
    X = w.draw(rng, N, 6)
    X[0] = w.init; X[-1] = w.goal_center
    w.X = X
    return w


def north_star_biased(N=1_000_000, goal_bias=5e-4):
    """The north star's world with the sample sets the LIBRARY's sampler draws (mpfmt_sample_free_biased, src/sampling.jl:11-45): free-space
    samples (nothing inside an obstacle) with a goal bias.  `device_sets(ctx, k)` draws set k on the device (the obstacle set must be
    uploaded); X here is a uniform placeholder until then.  goal_bias = 5e-4 puts ~500 of the 1e6 samples inside the goal ball (radius
    0.15 < r): columns of ~600 entries -- what a stored graph allows.  (A bias of 0.3 would put 3e5 samples inside one r-ball: 9e10
    edges, no CSC of that exists on any machine.)"""
    w = make("ns_r6_n1m_m200_goalbias", N, 6, 200, 0.10, 0.20, seed=3)
    w.goal_bias = goal_bias

    def device_set(ctx, k):
        X, _ = ctx.sample_free(1000 + k, w.N, init=w.init, goal_kind=1, goal_params=w.goal_params(), goal_ct=1, goal_bias=goal_bias)      # (1 = GOAL_BALL)
        return X
    w.device_set = device_set
    return w


BY_NAME = {"cfg1": cfg1, "cfg2": cfg2, "north_star": north_star, "cfg3": cfg3, "ns_clustered": north_star_clustered,
           "ns_biased": north_star_biased}


@dataclass
class DIWorkload:
    name: str
    X: np.ndarray          # (N, 2m) states (p, v)
    lohi: np.ndarray       # (M, 2, m) workspace boxes
    rho: float
    r: float               # cost radius
    ss_lo: np.ndarray
    ss_hi: np.ndarray

    @property
    def N(self):
        return self.X.shape[0]


def cfg4(N=100_000, m=2, M=20, vmax=0.5, rho=1.0, r=1.0, seed=5):
    """Kinodynamic FMT*: DoubleIntegrator(2; vmax=0.5, r=1.) (linearquadratic.jl:46-53), state R^4, cost radius 1
    (docs/MotionPlanning.ipynb cell 8), 20 2-D boxes.  BASELINE.json configs[3]."""
    rng = Stream(seed)
    init = np.concatenate([np.full(m, 0.1), np.zeros(m)])
    goal = np.concatenate([np.full(m, 0.9), np.zeros(m)])
    lohi = make_boxes(rng, M, m, 0.02, 0.08, [init[:m], goal[:m]])
    X = np.concatenate([rng.random((N, m)), vmax * (2 * rng.random((N, m)) - 1)], axis=1)
    X[0] = init
    X[-1] = goal
    return DIWorkload("cfg4_di_r4_n%d" % N, X, lohi, rho, r, np.concatenate([np.zeros(m), -vmax * np.ones(m)]),
                      np.concatenate([np.ones(m), vmax * np.ones(m)]))
