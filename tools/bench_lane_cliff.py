import os, sys, time
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
import numpy as np, torch
import geosss_amd as gs
n, steps = 1_000_000, 500
for d in (10, 11, 12, 16, 20):
    modes = gs.sample_sphere(d - 1, 3, seed=1234, rng="numpy")
    for label, pdf in ((f"vmf d={d} K=3", gs.MixtureModel([gs.VonMisesFisher(100.0 * m) for m in modes])),
                       (f"bingham d={d} eigen", gs.random_bingham(d=d, vmax=30.0, vmin=0.0, eigensystem=True, seed=6982))):
        x0 = gs.sample_sphere_device(d - 1, n, seed=1).T
        s = gs.ShrinkageSphericalSliceSampler(pdf, x0, 3521, mode="fast", placement="packed")
        s.advance(50)
        best = 1e9
        for _ in range(2):
            torch.cuda.synchronize(); t0 = time.perf_counter(); s.advance(steps); torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
        print(f"{label:22s} {n*steps/best:.3e}  {s._lib.gsss_kernel_name(s._target_dev.handle, 1, 0, 1).decode()}", flush=True)
