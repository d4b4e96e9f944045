#!/usr/bin/env python3
"""vMF mixtures, 10^6 chains x 500 steps: the library's packing against the one-chain-per-lane build (GSSS_ONE_PER_LANE=2 with a
library built with -DGSSS_VMF_ONE_ALL=1) over d and K.   GSSS_HIP_LIB=.../libgsss_oneall.so python tools/bench_vmf_pure_one.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import geosss_amd as gs
n, steps = 1_000_000, 500
for d in (3, 4, 5, 6, 8, 10):
    for K in (3, 4, 6, 16):
        modes = gs.sample_sphere(d - 1, K, seed=1234, rng="numpy")
        pdf = gs.MixtureModel([gs.VonMisesFisher(100.0 * m) for m in modes])
        x0 = gs.sample_sphere_device(d - 1, n, seed=1).T
        out = []
        for env in (None, "2"):
            if env is None:
                os.environ.pop("GSSS_ONE_PER_LANE", None)
            else:
                os.environ["GSSS_ONE_PER_LANE"] = env
            s = gs.ShrinkageSphericalSliceSampler(pdf, x0, 3521, mode="fast", placement="packed")
            s.advance(50)
            best = 1e9
            for _ in range(3):
                torch.cuda.synchronize(); t0 = time.perf_counter(); s.advance(steps); torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
            out.append(n * steps / best)
        os.environ.pop("GSSS_ONE_PER_LANE", None)
        print(f"vmf d={d:2d} K={K:2d}: default {out[0]:.3e}  one-per-lane build {out[1]:.3e} ({out[1] / out[0] - 1:+.1%})", flush=True)
