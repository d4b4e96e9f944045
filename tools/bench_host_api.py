#!/usr/bin/env python3
"""End-to-end rate of the host-facing call `sampler.sample(n)` -> numpy (includes the layout change and
the PCIe copy of every retained draw), next to the device-resident rate.  DESIGN.md "Measurement"."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import geosss_amd as gs

mus = 80.0 * np.array([[0.87, -0.37, 0.33], [-0.20, -0.89, -0.40], [0.19, 0.22, -0.96]])
pdf = gs.MixtureModel([gs.VonMisesFisher(m) for m in mus])
n = 1_000_000
x0 = gs.sample_sphere(2, n, seed=0)
for draws, thin in ((100, 1), (100, 10), (10, 100)):
    s = gs.ShrinkageSphericalSliceSampler(pdf, x0, seed=1)
    s.advance(50); torch.cuda.synchronize()
    t0 = time.perf_counter(); out = s.sample(draws, thin=thin); t1 = time.perf_counter()
    steps = (draws - 1) * thin
    s2 = gs.ShrinkageSphericalSliceSampler(pdf, x0, seed=1)
    s2.advance(50); torch.cuda.synchronize()
    t2 = time.perf_counter(); o2 = s2.sample(draws, thin=thin, as_tensor=True); torch.cuda.synchronize(); t3 = time.perf_counter()
    print(f"sample({draws}, thin={thin}) x {n} chains: numpy out {out.nbytes/1e9:.2f} GB in {t1-t0:.3f} s = {n*steps/(t1-t0):.3e} chain-steps/s"
          f" | device tensor in {t3-t2:.3f} s = {n*steps/(t3-t2):.3e} chain-steps/s")
