#!/usr/bin/env python3
"""Curve-vMF throughput over the dimension (the reference's sweep is d = 3 .. 24, sh/submit_job_curve_varying_ndim.sh:11;
BASELINE adds 50 and 200): 10^5 chains x 1000 steps per launch, default fast kernel against the all-double variant.
GPU box: python tools/bench_curve_sweep.py"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import geosss_amd as gs  # noqa: E402

n = 100_000
for d in [int(a) for a in sys.argv[1:]] or (3, 6, 9, 10, 12, 16, 17, 24, 32, 50, 64, 65, 100, 200):
    pdf = gs.CurvedVonMisesFisher(gs.SlerpCurve(gs.brownian_curve(10, d, 0.5, seed=4562)), 800.0)
    x0 = gs.sample_sphere_device(d - 1, n, seed=1).T
    row = []
    for screen in (True, False):
        s = gs.ShrinkageSphericalSliceSampler(pdf, x0, 3521, mode="fast", placement="packed", screen=screen)
        name = s._lib.gsss_kernel_name(s._target_dev.handle, 1, 0 if screen else 100, 1).decode()
        s.advance(200)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        s.advance(1000)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        row.append(f"{n * 1000 / dt:.3e} ({name})")
    print(f"d = {d:3d}: " + "   all-double: ".join(row), flush=True)
