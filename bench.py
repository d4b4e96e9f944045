#!/usr/bin/env python3
"""Benchmark of the many-chain geodesic shrinkage slice sampler on MI355X.

    python bench.py --gpus N --steps K --warmup W
    (N > 1: python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...)

Workload (BASELINE.json configs[1]): the README 3-component vMF mixture on S^2 (kappa = 80),
10^6 independent chains PER GPU (weak scaling; chains of rank r have ids r*10^6 ...), shrinkage
sampler.  One bench "step" = one launch of the sampler kernel advancing every chain by
`--inner` (default 100) MCMC transitions, keeping one thinned sample per launch; chain states
are resident in HBM before the timed region starts.  `value` = MCMC chain-steps per second over
all GPUs.  Rank 0 prints ONE JSON line (see DESIGN.md "Measurement" for every field).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0       # MI355X_MICROARCH.md: 8.0 TB/s spec
FP64_VALU_PEAK_TF = 78.6    # vendor vector-FP64 figure

README_MUS = 80.0 * np.array([[0.87, -0.37, 0.33], [-0.20, -0.89, -0.40], [0.19, 0.22, -0.96]])


def make_target(gs, name):
    if name == "vmfmix_readme":
        return gs.MixtureModel([gs.VonMisesFisher(m) for m in README_MUS]), 3
    if name == "vmfmix_k10_kappa500":
        import torch  # noqa: F401
        modes = gs.sample_sphere(2, 10, seed=1234)
        return gs.MixtureModel([gs.VonMisesFisher(500.0 * m) for m in modes]), 3
    if name == "bingham_d10":
        return gs.random_bingham(d=10, vmax=30.0, vmin=0.0, eigensystem=True, seed=6982), 10
    if name.startswith("curve_d"):
        d = int(name[len("curve_d"):])
        return gs.CurvedVonMisesFisher(gs.SlerpCurve(gs.brownian_curve(10, d, 0.5, seed=4562)), 800.0), d
    raise ValueError(name)


def oracle_target(orc, name):
    if name == "vmfmix_readme":
        return orc.Target.vmf_mixture(README_MUS)
    return None


def cpu_baseline(workload, d, budget_s=12.0):
    """The CPU oracle (C restatement of the reference loop, oracle/gsss_oracle.c) timed on the
    host cores on a bounded sample of the same workload."""
    from oracle import oracle as orc
    tgt = oracle_target(orc, workload)
    if tgt is None:
        return None
    cores = max(1, min(os.cpu_count() or 1, len(os.sched_getaffinity(0))))
    x0 = orc.sample_sphere(0, 4096, d)
    t0 = time.perf_counter()
    orc.run(tgt, x0, 20, seed=3521, keep_samples=False, n_threads=cores)
    rate = 4096 * 20 / (time.perf_counter() - t0)
    n_chains = 4096 * cores
    n_steps = int(max(20, min(2000, budget_s * rate / n_chains)))
    x0 = orc.sample_sphere(0, n_chains, d)
    t0 = time.perf_counter()
    out = orc.run(tgt, x0, n_steps, seed=3521, keep_samples=False, n_threads=cores)
    dt = time.perf_counter() - t0
    return {"value": n_chains * n_steps / dt, "unit": "chain-steps/s", "cores": cores, "kind": "port",
            "sample": f"{n_chains} chains x {n_steps} steps of the same target, OpenMP over chains, {dt:.1f} s",
            "tries_per_step": float(out["n_tries"].sum() / (n_chains * n_steps))}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--chains", type=int, default=1_000_000, help="chains per GPU")
    ap.add_argument("--inner", type=int, default=100, help="MCMC transitions per launch (= per bench step)")
    ap.add_argument("--workload", default="vmfmix_readme")
    ap.add_argument("--mode", default="exact")
    ap.add_argument("--variant", type=int, default=0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node N for --gpus N > 1")
    assert torch.cuda.is_available(), "bench.py needs a GPU"
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    import geosss_amd as gs
    from geosss_amd.ensemble import gather_states

    pdf, d = make_target(gs, args.workload)
    n = args.chains
    chain_offset = rank * n
    x0 = gs.sample_sphere_device(d - 1, n, seed=0, chain_offset=chain_offset)  # [d, n] on device
    sampler = gs.ShrinkageSphericalSliceSampler(pdf, x0.T, seed=3521, chain_offset=chain_offset, mode=args.mode,
                                                variant=args.variant)
    S = args.inner
    kept = torch.empty((1, d, n), dtype=torch.float64, device="cuda")

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        sampler.advance(S, thin=S, out=kept)
    tries0 = int(sampler._n_tries.sum().item())
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    barrier()
    t0 = time.perf_counter()
    for a, b in ev:
        a.record()
        sampler.advance(S, thin=S, out=kept)
        b.record()
    final = gather_states(sampler.state_device) if world > 1 else sampler.state_device
    barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())
    assert final.shape[1] == n * world

    kern_ms = float(np.mean([a.elapsed_time(b) for a, b in ev]))
    tries = int(sampler._n_tries.sum().item()) - tries0
    bad = int((sampler._err != 0).sum().item())
    chain_steps = n * S * args.steps
    if world > 1:
        agg = torch.tensor([tries, bad], dtype=torch.int64, device="cuda")
        dist.all_reduce(agg)
        tries, bad = int(agg[0].item()), int(agg[1].item())
    total_steps = chain_steps * world
    value = total_steps / elapsed

    if rank == 0:
        # algorithmic HBM bytes per chain-step: retained sample 8d/thin + state
        # load/store 16d/S + counters (3 x 8-byte read-modify-write words + err) / S
        bytes_per_step = 8.0 * d / S + (16.0 * d + 48.0) / S
        bytes_per_launch = bytes_per_step * n * S
        achieved = bytes_per_launch / (kern_ms * 1e-3) / 1e9
        lib = gs._lib.load()
        out = {
            "metric": "mcmc_chain_steps_per_sec", "value": value, "unit": "chain-steps/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"{args.workload}: shrinkage slice sampler, {n} chains/GPU x {S} transitions per "
                                   "launch, thin=%d, Philox4x32-10 stream" % S,
                       "target": args.workload, "d": d, "chains_per_gpu": n, "transitions_per_step": S,
                       "mode": args.mode,
                       "kernel": lib.gsss_variant_name(sampler._target_dev.handle, 0, args.variant).decode(),
                       "sharding": f"{world} x independent chain blocks, final states all-gathered" if world > 1
                       else "single GPU"},
            "tries_per_step": tries / total_steps, "chains_in_error": bad,
            "kernel_ms": kern_ms,
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": None,
                         "note": "state stays in registers across the launch; the path is FP64-VALU/transcendental "
                                 "bound, see DESIGN.md"},
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args.workload, d)
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
