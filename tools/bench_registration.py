#!/usr/bin/env python3
"""Throughput of the slice sampler on the protein registration target (scripts/protein_reg3d3d.py's model: 214 + 214
points, k = 20, sigma = 1, omega = 0.4) for many chains.  GPU box: python tools/bench_registration.py"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402

import geosss_amd as gs  # noqa: E402
from helpers import product_target  # noqa: E402

z = np.load(os.path.join(ROOT, "tests", "golden", "traj_cpd_protein.npz"))
pdf = product_target(z)
for n in (4096, 65536, 262144):
    x0 = gs.sample_sphere_device(3, n, seed=1).T
    s = gs.ShrinkageSphericalSliceSampler(pdf, x0, 5)
    s.advance(2)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    steps = 10
    s.advance(steps)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    tries = float(s._n_tries.sum().item()) / (n * (steps + 2))
    print(f"{n} chains: {n * steps / dt:.3e} chain-steps/s, {tries:.2f} tries/step, "
          f"{n * steps * (tries + 1) / dt:.3e} log_prob evaluations/s")
