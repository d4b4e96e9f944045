#!/usr/bin/env python3
"""Batched log-density on the device (`pdf.log_prob(tensor)`, gsss_logprob): points/s and the HBM rate it implies
(8 (d + 1) bytes per point: the point in, the value out).  What the reference's scripts call after sampling
(`logprob = pdf.log_prob(samples)`, scripts/curve_vMF.py:123-124).  GPU box: python tools/bench_logprob.py"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import geosss_amd as gs  # noqa: E402


def run(label, pdf, d, n):
    x = gs.sample_sphere_device(d - 1, n, seed=1).T.contiguous()
    pdf.log_prob(x[:1000])
    best = float("inf")
    for _ in range(3):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        out = pdf.log_prob(x)
        torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t0)
    print(f"{label:28s} n = {n:>11,d}  {n / best:.3e} points/s  {8 * (d + 1) * n / best / 1e9:8.1f} GB/s  ({best * 1e3:.2f} ms)", flush=True)
    return out


mus = np.array([[0.87, -0.37, 0.33], [-0.20, -0.89, -0.40], [0.19, 0.22, -0.96]])
run("vMF mixture d=3 K=3", gs.MixtureModel([gs.VonMisesFisher(80.0 * m) for m in mus]), 3, 100_000_000)
run("vMF mixture d=3 K=10", gs.MixtureModel([gs.VonMisesFisher(m) for m in 500 * gs.sample_sphere(2, 10, seed=1234)]), 3, 100_000_000)
run("Bingham d=10", gs.random_bingham(10, vmax=30.0, vmin=0.0, eigensystem=True, seed=6982), 10, 30_000_000)
for d, n in ((10, 30_000_000), (50, 5_000_000), (200, 1_000_000)):
    run(f"curve-vMF d={d}", gs.CurvedVonMisesFisher(gs.SlerpCurve(gs.brownian_curve(10, d, 0.5, seed=4562)), 800.0), d, n)
mus = 100.0 * gs.sample_sphere(49, 5, seed=1234)
run("vMF mixture d=50 K=5", gs.MixtureModel([gs.VonMisesFisher(m) for m in mus]), 50, 5_000_000)
