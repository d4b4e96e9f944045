#!/usr/bin/env python3
"""Host cost of one gsss_run launch through the Python class (advance(1) in a loop): what a caller who steps the sampler
one transition at a time pays per call."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import geosss_amd as gs
from bench import make_target

pdf, d = make_target(gs, "vmfmix_readme")
for n, placement in ((1, "auto"), (4096, "packed"), (100_000, "packed")):
    x0 = gs.sample_sphere_device(d - 1, n, seed=1).T
    s = gs.ShrinkageSphericalSliceSampler(pdf, x0, 3521, mode="fast", placement=placement)
    s.advance(10); torch.cuda.synchronize()
    reps = 2000
    t0 = time.perf_counter()
    for _ in range(reps):
        s.advance(1)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"n={n:7d} {placement:6s}: {dt / reps * 1e6:.1f} us per advance(1)")
