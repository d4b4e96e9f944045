#!/usr/bin/env python3
"""The rejection slice sampler (geoSSS reject, mcmc.py:340-375) on the bench targets beside the shrinkage sampler: chain-steps/s,
tries per step, kernel.  Best of three launches.  GPU box: python tools/bench_reject.py"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import geosss_amd as gs  # noqa: E402


def run(label, cls, pdf, d, n, steps):
    x0 = gs.sample_sphere_device(d - 1, n, seed=1).T
    s = cls(pdf, x0, 3521)
    s.advance(steps // 4)
    best = float("inf")
    for _ in range(3):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        s.advance(steps)
        torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t0)
    tries = float(s._n_tries.sum().item()) / (n * (3 * steps + steps // 4))
    name = s._lib.gsss_kernel_name(s._target_dev.handle, 1 if s.mode == "fast" else 0, 0, 1).decode()
    print(f"{label:30s} {n * steps / best:.3e} chain-steps/s  {tries:6.2f} tries/step  {n * steps * tries / best:.3e} tries/s  ({name})", flush=True)


mus = np.array([[0.87, -0.37, 0.33], [-0.20, -0.89, -0.40], [0.19, 0.22, -0.96]])
targets = [("README mixture", gs.MixtureModel([gs.VonMisesFisher(80.0 * m) for m in mus]), 3, 1_000_000, 400),
           ("K=10 mixture", gs.MixtureModel([gs.VonMisesFisher(m) for m in 500 * gs.sample_sphere(2, 10, seed=1234)]), 3, 1_000_000, 200),
           ("Bingham d=10", gs.random_bingham(10, vmax=30.0, vmin=0.0, eigensystem=True, seed=6982), 10, 1_000_000, 200),
           ("curve d=10", gs.CurvedVonMisesFisher(gs.SlerpCurve(gs.brownian_curve(10, 10, 0.5, seed=4562)), 800.0), 10, 100_000, 200),
           ("curve d=50", gs.CurvedVonMisesFisher(gs.SlerpCurve(gs.brownian_curve(10, 50, 0.5, seed=4562)), 800.0), 50, 100_000, 100)]
for name, pdf, d, n, steps in targets:
    run(f"shrink {name}", gs.ShrinkageSphericalSliceSampler, pdf, d, n, steps)
    run(f"reject {name}", gs.RejectionSphericalSliceSampler, pdf, d, n, steps)
