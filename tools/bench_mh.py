#!/usr/bin/env python3
"""Throughput of the Metropolis-Hastings family (RWMH, independence, mixture, spherical HMC with 10 leapfrog steps:
mh_kernel, gsss_mh.h) on the targets the paper compares the slice samplers with, many chains, best of three launches.
GPU box: python tools/bench_mh.py"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import geosss_amd as gs  # noqa: E402


def run(label, make, d, n, steps):
    x0 = gs.sample_sphere_device(d - 1, n, seed=1).T
    s = make(x0)
    s.reset(steps // 5)
    s.advance(steps // 5)
    best = float("inf")
    for _ in range(3):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        s.advance(steps)
        torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t0)
    acc = float(np.mean(s.n_accept_per_chain)) / (steps * 3 + steps // 5)
    print(f"{label:34s} {n:>9,d} chains  {n * steps / best:.3e} chain-steps/s  accept {acc:.2f}", flush=True)


mus = np.array([[0.87, -0.37, 0.33], [-0.20, -0.89, -0.40], [0.19, 0.22, -0.96]])
targets = [("README mixture d=3", gs.MixtureModel([gs.VonMisesFisher(80.0 * m) for m in mus]), 3, 1_000_000, 500),
           ("Bingham d=10", gs.random_bingham(10, vmax=30.0, vmin=0.0, eigensystem=True, seed=6982), 10, 1_000_000, 200),
           ("curve-vMF d=10", gs.CurvedVonMisesFisher(gs.SlerpCurve(gs.brownian_curve(10, 10, 0.5, seed=4562)), 800.0), 10, 100_000, 200),
           ("curve-vMF d=24", gs.CurvedVonMisesFisher(gs.SlerpCurve(gs.brownian_curve(10, 24, 0.5, seed=4562)), 800.0), 24, 100_000, 100)]
for name, pdf, d, n, steps in targets:
    run(f"rwmh  {name}", lambda x0: gs.MetropolisHastings(pdf, x0, 5, stepsize=0.1), d, n, steps)
    run(f"indep {name}", lambda x0: gs.IndependenceSampler(pdf, x0, 5), d, n, steps)
    run(f"mix   {name}", lambda x0: gs.MixtureRWMHIndependenceSampler(pdf, x0, 5, stepsize=0.1), d, n, steps)
    run(f"hmc   {name}", lambda x0: gs.SphericalHMC(pdf, x0, 5, stepsize=0.05, n_steps=10), d, n, max(steps // 5, 20))
