#!/usr/bin/env python3
"""Throughput of the cooperative (several lanes per chain) kernels over shapes no bench line covers: Bingham (eigenbasis
and dense) and vMF mixtures above d = 16, the all-double curve kernels, the exact mode.  10^5 chains x `steps` steps per
launch (fewer for the dense matrices).  GPU box:

    python tools/bench_shapes.py                  # GSSS_HIP_LIB=... for a side-by-side build
"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import geosss_amd as gs  # noqa: E402

n = 100_000


def run(label, pdf, d, steps, **kw):
    x0 = gs.sample_sphere_device(d - 1, n, seed=1).T
    s = gs.ShrinkageSphericalSliceSampler(pdf, x0, 3521, **kw)
    s.advance(max(steps // 5, 10))
    dt = float("inf")
    for _ in range(3):                                   # best of three: short launches, the clocks settle during the first
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        s.advance(steps)
        torch.cuda.synchronize()
        dt = min(dt, time.perf_counter() - t0)
    screen = kw.get("screen", True)
    name = s._lib.gsss_kernel_name(s._target_dev.handle, 1 if s.mode == "fast" else 0, 0 if screen else 100, 1).decode()
    print(f"{label:32s} {n * steps / dt:.3e} chain-steps/s  ({s.mode}: {name})", flush=True)


for d in (20, 32, 50, 100):
    run(f"bingham eigenbasis d={d}", gs.random_bingham(d, vmax=30.0, vmin=0.0, eigensystem=True, seed=6982), d, 500, mode="fast")
    run(f"bingham dense d={d}", gs.random_bingham(d, vmax=30.0, vmin=0.0, seed=6982), d, 200, mode="fast")
for d, K in ((20, 5), (32, 5), (50, 5), (64, 3), (100, 5), (128, 10), (200, 3)):
    mus = 100.0 * gs.sample_sphere(d - 1, K, seed=1234)
    run(f"vmf mixture d={d} K={K}", gs.MixtureModel([gs.VonMisesFisher(m) for m in mus]), d, 500, mode="fast")
for d in (50, 200):
    pdf = gs.CurvedVonMisesFisher(gs.SlerpCurve(gs.brownian_curve(10, d, 0.5, seed=4562)), 800.0)
    run(f"curve all-double d={d}", pdf, d, 500, mode="fast", screen=False)
    run(f"curve exact d={d}", pdf, d, 100, mode="exact")
for d in (50, 100, 200):
    mus = 100.0 * gs.sample_sphere(d - 1, 5, seed=1234)
    run(f"vmf mixture exact d={d} K=5", gs.MixtureModel([gs.VonMisesFisher(m) for m in mus]), d, 100, mode="exact")
run("bingham exact d=50", gs.random_bingham(50, vmax=30.0, vmin=0.0, eigensystem=True, seed=6982), 50, 100, mode="exact")
