#!/usr/bin/env python3
"""Lane kernels at 10^6 chains x 1000 steps over their shapes: two chains per lane (GSSS_ONE_PER_LANE=0) against one per lane
(=2, launched without the LDS of parked chains: more workgroups per CU where that LDS binds) and the library's choice."""
import os, sys, time, subprocess
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

if len(sys.argv) > 1:
    import torch
    import geosss_amd as gs
    n, steps = 1_000_000, 1000
    shapes = []
    for d in (3, 4, 6, 8, 10):
        for K in (3, 5, 10):
            modes = gs.sample_sphere(d - 1, K, seed=1234, rng="numpy")
            shapes.append((f"vmf_d{d}_K{K}", gs.MixtureModel([gs.VonMisesFisher(100.0 * m) for m in modes]), d))
        for eig in (True, False):
            shapes.append((f"bingham_d{d}_{'eigen' if eig else 'dense'}", gs.random_bingham(d=d, vmax=30.0, vmin=0.0, eigensystem=eig, seed=6982), d))
    for name, pdf, d in shapes:
        x0 = gs.sample_sphere_device(d - 1, n, seed=1).T
        s = gs.ShrinkageSphericalSliceSampler(pdf, x0, 3521, mode="fast", placement="packed")
        s.advance(100)
        best = 1e9
        for _ in range(2):
            torch.cuda.synchronize(); t0 = time.perf_counter(); s.advance(steps); torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
        print(f"{name} {n} {n * steps / best:.4e}", flush=True)
else:
    res = {}
    for mode in ("0", "2", "1"):
        env = dict(os.environ, GSSS_ONE_PER_LANE=mode)
        out = subprocess.run([sys.executable, __file__, "child"], env=env, capture_output=True, text=True).stdout
        for ln in out.splitlines():
            p = ln.split()
            if len(p) == 3 and p[1].isdigit():
                res[(p[0], mode)] = float(p[2])
    for name in [k[0] for k in res if k[1] == "0"]:
        a, b, c = res[(name, "0")], res.get((name, "2"), 0.0), res.get((name, "1"), 0.0)
        print(f"{name:22s}: two {a:.3e}  one {b:.3e}  library {c:.3e}  {'one wins' if b > 1.02 * a else ('two wins' if a > 1.02 * b else '')}"
              f"{'   <-- library picks the slower' if c < 0.97 * max(a, b) else ''}")
