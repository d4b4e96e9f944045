#!/usr/bin/env python3
"""numpy's stream on LARGE ensembles (one default_rng per chain, SeedSequence.spawn): chain-steps/s of the exact kernels against
the lane-per-chain fast kernel that reads the same stream (fast_kernel<..., NUMPY>), the Philox throughput kernel beside them."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import geosss_amd as gs
from bench import make_target

n, steps = int(os.environ.get("CHAINS", 100_000)), 1000
for name in ("vmfmix_readme", "bingham_d10", "curve_d10"):
    pdf, d = make_target(gs, name)[:2]
    x0 = gs.sample_sphere_device(d - 1, n, seed=1).T
    t0 = time.perf_counter()
    root = np.random.SeedSequence(7)
    for rng, mode in (("numpy", "exact"), ("numpy", "fast"), ("philox", "fast")):
        try:
            s = gs.ShrinkageSphericalSliceSampler(pdf, x0, root if rng == "numpy" else 7, rng=rng, mode=mode, placement="packed")
        except ValueError as e:
            print(f"{name:14s} rng={rng:6s} mode={mode:5s}: {e}")
            continue
        s.advance(50); torch.cuda.synchronize()
        t0 = time.perf_counter(); s.advance(steps); torch.cuda.synchronize(); dt = time.perf_counter() - t0
        print(f"{name:14s} rng={rng:6s} mode={mode:5s}: {n * steps / dt:.3e} chain-steps/s ({dt * 1e3:.1f} ms per {steps} steps of {n} chains)")
