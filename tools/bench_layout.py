#!/usr/bin/env python3
"""Bandwidth of the layout kernels (HBM-bound: 16 B of traffic per element moved)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import geosss_amd as gs

lib = gs._lib.load()
for n, rows, d in ((1_000_000, 100, 3), (1_000_000, 30, 10), (100_000, 20, 200), (4_000_000, 1, 3)):
    src = torch.randn(rows, d, n, dtype=torch.float64, device="cuda")
    dst = torch.empty(n, rows, d, dtype=torch.float64, device="cuda")
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for it in range(3):
        ev0.record()
        gs._lib.check(lib.gsss_samples_to_chains(src.data_ptr(), dst.data_ptr(), n, rows, d, 0,
                                                 torch.cuda.current_stream().cuda_stream))
        ev1.record()
        torch.cuda.synchronize()
    ms = ev0.elapsed_time(ev1)
    gb = 2 * src.numel() * 8 / 1e9
    ok = torch.equal(dst, src.permute(2, 0, 1).contiguous())
    print(f"samples_to_chains n={n} rows={rows} d={d}: {ms:.3f} ms, {gb/ms*1e3:.0f} GB/s, correct={ok}")
