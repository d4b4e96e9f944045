#!/usr/bin/env python3
"""Lane-per-chain shapes, 10^6 chains x 500 steps: the default (screened) kernel against the all-double one (screen=False) for
vMF mixtures (d = 3 .. 10, K = 3 / 5 / 10) and Bingham targets (d = 3 .. 10, eigenbasis and dense) -- does the dispatch pick
the faster kernel everywhere?"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import geosss_amd as gs

n, steps = 1_000_000, 500


def rate(pdf, d, **kw):
    x0 = gs.sample_sphere_device(d - 1, n, seed=1).T
    s = gs.ShrinkageSphericalSliceSampler(pdf, x0, 3521, mode="fast", **kw)
    s.advance(50)
    best = 1e9
    for _ in range(2):
        torch.cuda.synchronize(); t0 = time.perf_counter(); s.advance(steps); torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
    name = s._lib.gsss_kernel_name(s._target_dev.handle, 1, 0 if kw.get("screen", True) else 100, 1).decode()
    return n * steps / best, name


dims = [int(a) for a in sys.argv[1:]] or list(range(3, 17))   # (round 4: lane kernels up to d = 16; screen=False is the cooperative kernel there)
for d in dims:
    for K in (3, 5, 10):
        modes = gs.sample_sphere(d - 1, K, seed=1234, rng="numpy")
        pdf = gs.MixtureModel([gs.VonMisesFisher(100.0 * m) for m in modes])
        a, na = rate(pdf, d)
        b, nb = rate(pdf, d, screen=False)
        print(f"vmf d={d:2d} K={K:2d}: {a:.3e} ({na})  all-double {b:.3e}  {'<-- all-double wins' if b > 1.03 * a else ''}", flush=True)
    for eig in (True, False):
        pdf = gs.random_bingham(d=d, vmax=30.0, vmin=0.0, eigensystem=eig, seed=6982)
        a, na = rate(pdf, d)
        b, nb = rate(pdf, d, screen=False)
        print(f"bingham d={d:2d} {'eigen' if eig else 'dense'}: {a:.3e} ({na})  all-double {b:.3e}  {'<-- all-double wins' if b > 1.03 * a else ''}", flush=True)
