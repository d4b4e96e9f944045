#!/usr/bin/env python3
"""The three experiments of the geoSSS paper, every chain of an ensemble at once: the metrics the reference's scripts
report (scripts/mixture_vMF.py, scripts/bingham.py, scripts/curve_vMF.py and their notebooks), per sampler, averaged
over `--chains` independent chains that run side by side on one MI355X.

    python examples/paper_experiments.py [--chains 256] [--draws 10000]

mixture   vMF mixture on S^9, K = 5, kappa = 100 (scripts/mixture_vMF.py): KL of the mode occupancy to the uniform weights,
          effective sample size of the first coordinate
bingham   Bingham on S^9, spectrum 0 .. 30 (scripts/bingham.py:123-134): hopping frequency across the mode's equator, ESS
curve     vMF around a curve on S^2 / S^9, kappa = 800 (scripts/curve_vMF.py): geodesic distance of consecutive draws, ESS,
          and on S^2 the KL divergence on the spiral grid (visualize_curve_vMF.ipynb)
"""
import argparse
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))   # run from a checkout
import geosss_amd as gs  # noqa: E402
from geosss_amd import diagnostics as dg  # noqa: E402

METHODS = {"sss-reject": gs.RejectionSphericalSliceSampler, "sss-shrink": gs.ShrinkageSphericalSliceSampler,
           "rwmh": gs.MetropolisHastings, "hmc": gs.SphericalHMC}


def chains(method, pdf, x0, n_draws, seed=3521):
    """(chains, draws, d) on the device, burn-in of 20 % dropped, and the seconds it took"""
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    x = METHODS[method](pdf, x0, seed).sample(n_draws, burnin=0.2, as_tensor=True)
    torch.cuda.synchronize()
    return x, time.perf_counter() - t0


def ess_first_coordinate(x):
    return float(dg.n_eff(x[:, :, 0].contiguous()).mean())


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--chains", type=int, default=256)
    ap.add_argument("--draws", type=int, default=10_000)
    a = ap.parse_args()
    n, m = a.chains, a.draws

    print(f"{n} chains x {m} draws per sampler\n")
    d, K, kappa = 10, 5, 100.0
    mus = kappa * gs.sample_sphere(d - 1, K, seed=1234)
    pdf = gs.MixtureModel([gs.VonMisesFisher(mu) for mu in mus])
    x0 = gs.sample_sphere(d - 1, n, seed=1)
    print(f"vMF mixture, d = {d}, K = {K}, kappa = {kappa:g}")
    for method in METHODS:
        x, dt = chains(method, pdf, x0, m)
        kl = np.mean([float(dg.mode_kl(dg.mode_occupancy(c, mus), np.full(K, 1.0 / K))) for c in x[: min(n, 64)]])
        print(f"  {method:10s} KL(mode occupancy || weights) {kl:8.4f}   ESS(x_1) {ess_first_coordinate(x):8.1f}   {dt:5.2f} s")

    pdf = gs.random_bingham(d, vmax=30.0, vmin=0.0, eigensystem=True, seed=6982)
    print(f"\nBingham, d = {d}, eigenvalues 0 .. 30")
    for method in METHODS:
        x, dt = chains(method, pdf, x0, m)
        hop = float(dg.hopping_frequency(x, pdf.mode).mean())
        print(f"  {method:10s} hopping frequency {hop:8.4f}   ESS(x_1) {ess_first_coordinate(x):8.1f}   {dt:5.2f} s")

    for d in (3, 10):
        pdf = gs.CurvedVonMisesFisher(gs.SlerpCurve(gs.brownian_curve(n_points=10, dimension=d, step_size=0.5, seed=4562)), 800.0)
        x0 = gs.sample_sphere(d - 1, n, seed=1345)
        print(f"\ncurve-vMF, d = {d}, kappa = 800")
        for method in METHODS:
            x, dt = chains(method, pdf, x0, m)
            step = float(dg.distance(x[:, 1:], x[:, :-1]).mean())
            line = f"  {method:10s} geodesic step {step:7.4f}   ESS(x_1) {ess_first_coordinate(x):8.1f}"
            if d == 3:
                line += f"   KL on the spiral grid {float(dg.grid_kl(pdf, x[: min(n, 16)]).mean()):7.4f}"
            print(line + f"   {dt:5.2f} s")


if __name__ == "__main__":
    main()
