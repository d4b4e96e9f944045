#!/usr/bin/env python3
"""Where does one wavefront per chain (spread placement, 8 speculative tries per step) stop beating the packed throughput
kernels?  chain-steps/s of both placements over the ensemble size, lane shapes."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import geosss_amd as gs
from bench import make_target

steps = 2000
for name in ("vmfmix_readme", "vmfmix_k10_kappa500", "bingham_d10", "curve_d10"):
    pdf, d = make_target(gs, name)
    for n in [int(v) for v in os.environ.get("SIZES", "1024,2048,4096,8192,16384,32768,65536").split(",")]:
        x0 = gs.sample_sphere_device(d - 1, n, seed=1).T
        out = {}
        for placement in ("packed", "spread") if n <= 65536 else ("packed",):
            s = gs.ShrinkageSphericalSliceSampler(pdf, x0, 3521, mode="fast", placement=placement)
            s.advance(100)
            best = 1e9
            for _ in range(2):
                torch.cuda.synchronize(); t0 = time.perf_counter(); s.advance(steps); torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
            out[placement] = n * steps / best
        sp = out.get("spread", 0.0)
        print(f"{name:20s} n={n:7d}: packed {out['packed']:.3e}  spread {sp:.3e}  {'spread wins' if sp > out['packed'] else ''}", flush=True)
