#!/usr/bin/env python3
"""Bingham targets, 10^6 chains x 500 steps: the library's choice of packing (two chains per lane at this size) against the
one-chain-per-lane build (GSSS_ONE_PER_LANE=2 -> screened_kernel<.., STAGE>, no code for a parked chain), with and without
chain-major retained rows (thin 100).  Decides do_screened_run's rule for large ensembles.   python tools/bench_pure_one.py [d ...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import geosss_amd as gs

n, steps = 1_000_000, 500
dims = [int(a) for a in sys.argv[1:]] or list(range(3, 11))


def rate(pdf, d, rows):
    x0 = gs.sample_sphere_device(d - 1, n, seed=1).T
    s = gs.ShrinkageSphericalSliceSampler(pdf, x0, 3521, mode="fast", placement="packed")
    out = torch.empty((n, steps // 100, d), dtype=torch.float64, device="cuda") if rows else None
    s.advance(50)
    best = 1e9
    for _ in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        if rows:
            s.advance(steps, thin=100, out=out, chain_major=True)
        else:
            s.advance(steps)
        torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
    return n * steps / best


for d in dims:
    for eig in (True, False):
        pdf = gs.random_bingham(d=d, vmax=30.0, vmin=0.0, eigensystem=eig, seed=6982)
        row = []
        for rows in (False, True):
            os.environ.pop("GSSS_ONE_PER_LANE", None)
            a = rate(pdf, d, rows)
            os.environ["GSSS_ONE_PER_LANE"] = "2"
            b = rate(pdf, d, rows)
            os.environ.pop("GSSS_ONE_PER_LANE", None)
            row.append(f"{'rows' if rows else 'no rows'}: default {a:.3e}  one per lane {b:.3e} ({b / a - 1:+.1%})")
        print(f"bingham d={d:2d} {'eigen' if eig else 'dense'}: " + "   ".join(row), flush=True)
