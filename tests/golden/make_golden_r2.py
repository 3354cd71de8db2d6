"""Round-2 golden fixtures (SURVEY.md 8c items iv and v), generated with the CPU oracle:
  di_pairs.npz       512 double-integrator state pairs -> (cost, t*) of steer (linearquadratic.jl:175-195) and the 5 collision
                     waypoints x(v, w, t*, s), s = linspace(0, t*, 5) (:85-88), with rho = 1, r = 1 (notebook cell 8)
  stream_heads.json  the first 16 uniforms of the workload stream for every seed the workloads use, plus the SplitMix64
                     known-answer vector (seed 1234567)
Run from the repo root:  python tests/golden/make_golden_r2.py
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from oracle import oracle as orc  # noqa: E402
import motionplanning_jl_amd as mp  # noqa: E402


def gen_di_pairs():
    st = mp.workloads.Stream(4040)
    n, m, vmax, rho, r = 512, 2, 0.5, 1.0, 1.0
    X0 = np.concatenate([st.random((n, m)), vmax * (2 * st.random((n, m)) - 1)], axis=1)
    # targets near enough that about half the pairs are within the cost radius
    X1 = X0.copy()
    X1[:, :m] += X0[:, m:] * 0.4 + 0.12 * (st.random((n, m)) - 0.5)          # roughly where the drift takes the state
    X1[:, m:] = np.clip(X0[:, m:] + 0.5 * (st.random((n, m)) - 0.5), -vmax, vmax)
    X1[5] = X0[5]                                                  # x0 == x1 -> (0, 0), linearquadratic.jl:192
    cost = np.empty(n); topt = np.empty(n); wps = np.empty((n, 5, 2 * m))
    for i in range(n):
        cost[i], topt[i] = orc.di_steer(X0[i], X1[i], rho, r)
        wps[i] = orc.di_waypoints(X0[i], X1[i], rho, r)
    np.savez(os.path.join(HERE, "di_pairs.npz"), X0=X0, X1=X1, rho=rho, r=r, cost=cost, topt=topt, waypoints=wps)
    print("di_pairs: %d pairs, %d within the cost radius" % (n, int((cost <= r).sum())))


def gen_stream_heads():
    out = {"splitmix64_seed_1234567_first5": [str(int(x)) for x in mp.workloads.splitmix64(1234567, 5)], "heads": {}}
    for seed in (1, 2, 3, 4, 5, 6):
        out["heads"][str(seed)] = [float.hex(float(u)) for u in mp.workloads.Stream(seed).random((16,))]
    json.dump(out, open(os.path.join(HERE, "stream_heads.json"), "w"), indent=1)


if __name__ == "__main__":
    gen_di_pairs()
    gen_stream_heads()
