#!/usr/bin/env python3
"""K = 12 and K = 16 component vMF mixtures on S^2 (the largest component bucket of the screened lane kernel), 10^6 chains."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import geosss_amd as gs
n, steps = 1_000_000, 1000
for K in (12, 16):
    modes = gs.sample_sphere(2, K, seed=1234, rng="numpy")
    pdf = gs.MixtureModel([gs.VonMisesFisher(500.0 * m) for m in modes])
    x0 = gs.sample_sphere_device(2, n, seed=1).T
    s = gs.ShrinkageSphericalSliceSampler(pdf, x0, 3521, mode="fast")
    s.advance(100)
    best = 1e9
    for _ in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter(); s.advance(steps); torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
    name = s._lib.gsss_kernel_name(s._target_dev.handle, 1, 0, 1).decode()
    print(f"K={K}: {n * steps / best:.3e} chain-steps/s, {best * 1e3:.1f} ms ({name})")
