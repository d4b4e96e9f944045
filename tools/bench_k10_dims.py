import os, sys, time
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
import torch
import geosss_amd as gs
n, steps = 1_000_000, 500
for d in (3, 4, 6, 8, 10):
    modes = gs.sample_sphere(d - 1, 10, seed=1234, rng="numpy")
    pdf = gs.MixtureModel([gs.VonMisesFisher(100.0 * m) for m in modes])
    x0 = gs.sample_sphere_device(d - 1, n, seed=1).T
    s = gs.ShrinkageSphericalSliceSampler(pdf, x0, 3521, mode="fast", placement="packed")
    s.advance(50)
    best = 1e9
    for _ in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter(); s.advance(steps); torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
    print(f"vmf K=10 d={d}: {n * steps / best:.3e}", flush=True)
