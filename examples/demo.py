#!/usr/bin/env python3
"""The reference's demo (demo.ipynb) with the import swapped: the four samplers on the README mixture of three von
Mises-Fisher components on S^2, first as the reference runs them (one chain, numpy's stream: the very numbers geosss prints),
then as this package is meant to be used (10^5 chains at once, diagnostics without stored draws).  Needs one MI355X.

    python examples/demo.py
"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))   # run from a checkout
import geosss_amd as gs  # noqa: E402

mus = np.array([[0.86981638, -0.37077248, 0.32549536],
                [-0.19772391, -0.89279985, -0.40473902],
                [0.19047726, 0.22240888, -0.95616562]])
pdf = gs.MixtureModel([gs.VonMisesFisher(80.0 * mu) for mu in mus])
init_state = np.array([-0.86333052, 0.18685286, -0.46877117])
n_samples, burnin, seed = 1000, 100, 3521

print("one chain, numpy's stream (the reference's own draws):")
samplers = {"sss-reject": gs.RejectionSphericalSliceSampler, "sss-shrink": gs.ShrinkageSphericalSliceSampler,
            "rwmh": gs.MetropolisHastings, "hmc": gs.SphericalHMC}
for name, cls in samplers.items():
    t0 = time.perf_counter()
    x = cls(pdf, init_state, seed, rng="numpy").sample(n_samples, burnin)
    dt = time.perf_counter() - t0
    occupancy = np.bincount(np.argmax(x @ mus.T, axis=1), minlength=3) / len(x)
    print(f"  {name:10s} {x.shape}  first draw {np.round(x[0], 8)}  mode occupancy {np.round(occupancy, 2)}  {dt * 1e3:.0f} ms")

print("10^5 chains at once (counter-based stream), running diagnostics instead of stored draws:")
x0 = gs.sample_sphere(2, 100_000, seed=0)
for name in ("sss-reject", "sss-shrink"):
    s = samplers[name](pdf, x0, seed)
    s.advance(burnin)
    torch.cuda.synchronize()
    s.enable_stats(lags=32, modes=mus)
    t0 = time.perf_counter()
    s.advance(n_samples, thin=1, keep=False)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    st = s.stats()
    occ = st["mode_occupancy"].mean(0).cpu().numpy()
    print(f"  {name:10s} {s.n_chains * n_samples / dt:.2e} chain-steps/s  mode occupancy {np.round(occ, 3)}  "
          f"IAT of x_1 {float(st['iat'].mean()):.1f}  rejections/step {s.n_reject / (s.n_chains * (n_samples + burnin)):.2f}")
