#!/usr/bin/env python3
"""End-to-end rate of the host-facing call `sampler.sample(n)` -> numpy (the reference's return value, geosss/mcmc.py:55-77: every
retained draw crosses PCIe), next to the device-resident rate (`as_tensor=True`).  DESIGN.md "Measurement"; never `value`.

    python tools/bench_host_api.py [--json gpurun_out/host_api.json]

cold: the first call of its size (page-locks the array's memory); warm: the previous array was dropped, its pages are reused;
one_piece: blocks=1 (one launch sequence, one copy); pageable: round 4's path (`out.cpu().numpy()`), for comparison."""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import geosss_amd as gs
from geosss_amd import _pinned

ap = argparse.ArgumentParser()
ap.add_argument("--json", default=None)
ap.add_argument("--chains", type=int, default=1_000_000)
args = ap.parse_args()

mus = 80.0 * np.array([[0.87, -0.37, 0.33], [-0.20, -0.89, -0.40], [0.19, 0.22, -0.96]])
pdf = gs.MixtureModel([gs.VonMisesFisher(m) for m in mus])
n = args.chains
x0 = gs.sample_sphere_device(2, n, seed=0).T.contiguous()


def timed(fn):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    out = fn()
    torch.cuda.synchronize()
    return time.perf_counter() - t0, out


rows = []
for draws, thin in ((100, 1), (100, 10), (10, 100)):
    steps = (draws - 1) * thin
    rec = {"call": f"sample({draws}, thin={thin})", "chains": n, "chain_steps": n * steps, "array_gb": 8e-9 * n * draws * 3}
    _pinned.trim()
    for label, kw in (("cold", {}), ("warm", {}), ("warm_again", {}), ("one_piece", {"blocks": 1})):
        s = gs.ShrinkageSphericalSliceSampler(pdf, x0, seed=1)
        s.advance(50)
        dt, out = timed(lambda: s.sample(draws, thin=thin, **kw))
        rec[label] = {"s": dt, "chain_steps_per_s": n * steps / dt, "gb_per_s": out.nbytes / dt / 1e9, "blocks": s._plan_blocks(0, draws, thin, kw.get("blocks"))}
        del out
    s = gs.ShrinkageSphericalSliceSampler(pdf, x0, seed=1)
    s.advance(50)
    dt, out = timed(lambda: s.sample(draws, thin=thin, as_tensor=True))
    rec["device_tensor"] = {"s": dt, "chain_steps_per_s": n * steps / dt}
    dt2, host = timed(lambda: out.cpu().numpy())
    rec["pageable"] = {"s": dt + dt2, "chain_steps_per_s": n * steps / (dt + dt2), "gb_per_s": host.nbytes / dt2 / 1e9}
    del out, host
    rows.append(rec)
    print(f"{rec['call']} x {n} chains ({rec['array_gb']:.2f} GB): " + "  ".join(
        f"{k} {rec[k]['s'] * 1e3:.1f} ms = {rec[k]['chain_steps_per_s']:.3e}/s" + (f" ({rec[k]['gb_per_s']:.1f} GB/s)" if "gb_per_s" in rec[k] else "")
        for k in ("cold", "warm", "warm_again", "one_piece", "device_tensor", "pageable")), flush=True)
if args.json:
    json.dump({"host_api": rows}, open(args.json, "w"), indent=1)
