#!/usr/bin/env python3
"""State-norm drift over long runs (the reference never renormalises its state, mcmc.py:396; neither do the kernels):
| |x| - 1 | after 10^6 transitions for the S^2 tangent stream (d = 3), the normal-vector stream (d = 10) and a group kernel."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

import geosss_amd as gs  # noqa: E402

mus = 80.0 * np.array([[0.87, -0.37, 0.33], [-0.20, -0.89, -0.40], [0.19, 0.22, -0.96]])
cases = {"vmf mixture d=3": gs.MixtureModel([gs.VonMisesFisher(m) for m in mus]),
         "bingham d=10": gs.random_bingham(d=10, vmax=30.0, vmin=0.0, eigensystem=True, seed=6982),
         "curve d=10": gs.CurvedVonMisesFisher(gs.SlerpCurve(gs.brownian_curve(10, 10, 0.5, seed=4562)), 800.0)}
for name, pdf in cases.items():
    s = gs.ShrinkageSphericalSliceSampler(pdf, gs.sample_sphere(pdf.d - 1, 4096, seed=1), 3, placement="packed")
    for _ in range(10):
        s.advance(100_000)
    nrm = np.linalg.norm(s.state, axis=1)
    print(f"{name}: after 1e6 steps max | |x| - 1 | = {np.max(np.abs(nrm - 1)):.2e}, errors {int((s.errors != 0).sum())}", flush=True)
