#!/usr/bin/env python3
"""Few long chains (the paper's own job shape: 10 chains x 10^6 steps, sh/parallelized_job_curve.sh):
latency-bound on a GPU -- one lane per chain.  Steps/s per chain for each stream and kernel family."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import geosss_amd as gs

mus = 80.0 * np.array([[0.87, -0.37, 0.33], [-0.20, -0.89, -0.40], [0.19, 0.22, -0.96]])
pdf = gs.MixtureModel([gs.VonMisesFisher(m) for m in mus])
for n in (1, 10):
    x0 = gs.sample_sphere(2, n, seed=0).reshape(n, 3)
    for rng, mode in (("numpy", "exact"), ("numpy", "fast"), ("philox", "exact"), ("philox", "fast")):
        s = gs.ShrinkageSphericalSliceSampler(pdf, x0 if n > 1 else x0[0], 3521, rng=rng, mode=mode)
        s.advance(1000); torch.cuda.synchronize()
        steps = 100_000
        t0 = time.perf_counter(); s.advance(steps); torch.cuda.synchronize(); dt = time.perf_counter() - t0
        print(f"{n:3d} chain(s) rng={rng:6s} mode={mode:5s}: {steps/dt:.3e} steps/s per chain ({n*steps/dt:.3e} total)")
