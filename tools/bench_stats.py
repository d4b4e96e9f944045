#!/usr/bin/env python3
"""One `Stats = true` launch shape of the headline target for profiling: 10^6 README chains, running statistics
on every state (thin = 1) with the lag sums on, nothing stored.  Prints kernel time per launch and the bytes the accumulators'
rows make per retained draw (read-modify-write in HBM: gsss_run_args.stats_dev).

    python tools/bench_stats.py [--chains N] [--steps S] [--thin T] [--lags L] [--launches K]
"""
import argparse, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import geosss_amd as gs
import bench

ap = argparse.ArgumentParser()
ap.add_argument("--chains", type=int, default=1_000_000)
ap.add_argument("--steps", type=int, default=200)
ap.add_argument("--thin", type=int, default=1)
ap.add_argument("--lags", type=int, default=32)
ap.add_argument("--launches", type=int, default=5)
ap.add_argument("--workload", default="vmfmix_readme")
a = ap.parse_args()

pdf, d = bench.make_target(gs, a.workload)
x0 = gs.sample_sphere_device(d - 1, a.chains, seed=0)
s = gs.ShrinkageSphericalSliceSampler(pdf, x0.T, seed=3521).enable_stats(lags=a.lags)
rows = int(s._stats["acc"].shape[0])
s.advance(100)
s.advance(a.steps, thin=a.thin, keep=False)          # warm-up
ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(a.launches)]
for e0, e1 in ev:
    e0.record()
    s.advance(a.steps, thin=a.thin, keep=False)
    e1.record()
torch.cuda.synchronize()
ms = float(np.mean([e0.elapsed_time(e1) for e0, e1 in ev]))
draws = a.chains * (a.steps // a.thin)
out = {"workload": a.workload, "chains": a.chains, "steps": a.steps, "thin": a.thin, "lags": a.lags, "kernel_ms": ms,
       "chain_steps_per_s": a.chains * a.steps / (ms * 1e-3), "stats_rows_per_chain": rows,
       "accumulator_bytes_per_chain": 8 * rows,
       # every retained draw reads and rewrites the rows it touches; an upper bound: all of them, both ways
       "rmw_bytes_per_launch_upper": 16.0 * rows * draws,
       "state_bytes_per_launch": (16.0 * d + 16.0) * a.chains,
       "n_eff_mean": float(s.stats()["n_eff"].mean().item())}
print(json.dumps(out))
