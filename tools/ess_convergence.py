#!/usr/bin/env python3
"""How long must a curve-vMF chain run before the between-chain estimator of tau (diagnostics.ess_between_chains) stops moving?
Doubling windows on a sub-ensemble: window k runs n_k = 2 n_{k-1} steps (the previous windows are its burn-in) and prints
tau_k; converged when n_k >= 20 tau_k and tau_k agrees with tau_{k-1}.   python tools/ess_convergence.py curve_d10 16384 [max_steps]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402
import geosss_amd as gs  # noqa: E402
from geosss_amd import diagnostics as dg  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "curve_d10"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 16384
max_steps = int(sys.argv[3]) if len(sys.argv) > 3 else 4_000_000
pdf, d = bench.make_target(gs, name)
x0 = gs.sample_sphere_device(d - 1, n, seed=0)
s = gs.ShrinkageSphericalSliceSampler(pdf, x0.T, seed=3521, placement="packed")
steps, total = 8192, 0
while total + steps <= max_steps:
    thin = max(1, steps // 512)
    s.enable_stats(lags=8, second_moment=False)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    s.advance(steps, thin=thin, keep=False)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    r = s.stats()
    bc = dg.ess_between_chains(r["proj_mean"], r["n"], r["proj_var"])
    total += steps
    print(f"{name} chains {n} window {steps} steps (after {total - steps} of burn-in) thin {thin}: tau = {bc['tau'] * thin:.0f} steps, "
          f"n / tau = {steps / (bc['tau'] * thin):.1f}, {dt:.2f} s, {n * steps / dt:.3e} chain-steps/s", flush=True)
    steps *= 2
