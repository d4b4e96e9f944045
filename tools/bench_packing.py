#!/usr/bin/env python3
"""Lane kernels: two chains per lane (GSSS_ONE_PER_LANE=0) against one per lane (=2) over the ensemble size, packed placement,
2000 steps per launch (partial rounds sliced): which packing wins where."""
import os, sys, time, subprocess
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

if len(sys.argv) > 1:
    import torch
    import geosss_amd as gs
    from bench import make_target
    steps = 2000
    for name in ("vmfmix_readme", "vmfmix_k10_kappa500", "bingham_d10"):
        pdf, d = make_target(gs, name)
        for n in range(131072, 1048577, 65536):
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
                res[(p[0], int(p[1]), mode)] = float(p[2])
    for (name, n, mode) in sorted(k for k in res if k[2] == "0"):
        a, b, c = res[(name, n, "0")], res.get((name, n, "2"), 0.0), res.get((name, n, "1"), 0.0)
        print(f"{name:20s} n={n:8d}: two {a:.3e}  one {b:.3e}  library {c:.3e}  {'one wins' if b > 1.02 * a else ('two wins' if a > 1.02 * b else '')}"
              f"{'   <-- library picks the slower' if c < 0.97 * max(a, b) else ''}")
