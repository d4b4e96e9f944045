"""The oracle at the launch shapes bench.py times.

The small same-stream parity tests (test_hip_parity.py) run a few thousand chains for ~100 steps: one resident round of
workgroups, no slicing.  The bench's launches are different animals -- 10^6 chains x 1000 steps of the README mixture run
1.5 rounds of five-workgroups-per-CU `screened_kernel<3, ScreenVmf<3,3>>`, two chains per lane, the last 674 chunks cut into
128-step slices that hand their state over through HBM; 10^5 chains of the d = 50 curve run `curvespec_kernel<4,3,10,+1>` with
EVERY chunk sliced.  Chains are keyed by their global id, so the oracle can check any subset of such a launch in seconds:
blocks of chain ids from the unsliced rounds, from the sliced tail, across the boundary between them and from the ragged
last chunk are compared with `oracle.run(..., chain_offset=...)` -- kept rows and final states at 1e-10, tries exactly.
(geosss/mcmc.py:382-401 for every chain)
"""
import ctypes as C
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

pytestmark = pytest.mark.gpu

TOL = 1e-10


@pytest.fixture(scope="module")
def gs():
    import torch
    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    import geosss_amd
    geosss_amd._lib.require_device()
    return geosss_amd


def _last_launch(gs):
    grid, steps, frac = C.c_int64(0), C.c_int32(0), C.c_double(0.0)
    gs._lib.load().gsss_last_launch(C.byref(grid), C.byref(steps), C.byref(frac))
    return int(grid.value), int(steps.value), float(frac.value)


def _check_blocks(gs, oracle, workload, n, blocks, kernel_prefix, chains_per_chunk, expect_sliced, monkeypatch, per_cu):
    """Two legs, NEITHER of which skips.  (1) The launch as this box's bench.py would time it (no environment overrides): whatever
    the library decides here -- sliced like round 4's box or not -- the same blocks of chain ids are held to the oracle.  (2) When
    leg 1 did not reproduce the bench shape of the profiled box (`per_cu` resident workgroups per CU, the last partial round or
    every chunk in 128-step slices), the launch again with that plan forced (GSSS_RESIDENT_PER_CU, GSSS_SLICE_STEPS=128), so
    that the sliced path is pinned on every box."""
    import torch
    import bench
    S, thin, seed = 1000, 100, 3521                                   # bench.py: --inner 1000 --thin 100, seed 3521
    pdf, d = bench.make_target(gs, workload)
    tgt = bench.oracle_target(oracle, gs, workload)
    x0 = gs.sample_sphere_device(d - 1, n, seed=0)                    # [d, n], as bench.py draws them
    x0_host = x0.T.contiguous().cpu().numpy()
    n_chunks = -(-n // chains_per_chunk)
    lo_f, hi_f = expect_sliced
    want_cache, legs = {}, []
    for leg in ("as this box launches it", "the profiled box's plan, forced"):
        for var in ("GSSS_SLICE_STEPS", "GSSS_ONE_PER_LANE", "GSSS_CURVE_L2", "GSSS_RESIDENT_PER_CU"):
            monkeypatch.delenv(var, raising=False)
        if leg.endswith("forced"):
            monkeypatch.setenv("GSSS_SLICE_STEPS", "128")
            monkeypatch.setenv("GSSS_RESIDENT_PER_CU", str(per_cu))
        s = gs.ShrinkageSphericalSliceSampler(pdf, x0.T, seed=seed)
        name = s._lib.gsss_kernel_name(s._target_dev.handle, 1, 0, 1).decode()
        assert s.mode == "fast" and name.startswith(kernel_prefix), name
        layout = bench.pick_layout("auto", name, d, thin)             # the layout of the kept rows bench.py's line is timed on
        kept, kw = bench.kept_buffer(torch, layout, n, S, thin, d)
        s.advance(S, thin=thin, out=kept, **kw)
        torch.cuda.synchronize()
        grid, slice_steps, frac = _last_launch(gs)
        bench_shape = slice_steps == 128 and lo_f <= frac <= hi_f
        if leg.endswith("forced"):
            assert bench_shape, (grid, slice_steps, frac)
        first_sliced = n_chunks - int(round(frac * n_chunks)) if slice_steps else n_chunks   # chunks below run the whole launch in one workgroup
        assert np.all(s.errors == 0)
        n_checked, where = 0, []
        for kind, lo, m in blocks(first_sliced, n_chunks):
            lo, m = max(0, min(lo, n - m)), min(m, n)
            ids = slice(lo, lo + m)
            if (lo, m) not in want_cache:
                want_cache[(lo, m)] = oracle.run(tgt, x0_host[ids], S, seed=seed, chain_offset=lo, thin=thin, n_threads=16)
            want = want_cache[(lo, m)]
            assert np.all(want["err"] == 0)
            got_rows = (kept[ids] if layout == "chains" else kept[:, :, ids].permute(2, 0, 1)).cpu().numpy()   # (chains, rows, d)
            assert np.array_equal(s._n_tries[ids].cpu().numpy(), want["n_tries"]), (leg, kind)
            assert np.array_equal(s._n_reject[ids].cpu().numpy(), want["n_reject"]), (leg, kind)
            assert np.max(np.abs(got_rows - want["samples"])) < TOL, (leg, kind)
            assert np.max(np.abs(s.state_device[:, ids].T.cpu().numpy() - want["state"])) < TOL, (leg, kind)
            n_checked += m
            where.append(kind)
        legs.append((leg, slice_steps, round(frac, 4), n_checked))
        del s, kept
        if bench_shape:
            break
    print(f"{workload}: " + "; ".join(f"{leg}: slice_steps {st}, sliced share {fr}, {nc} chains against the oracle" for leg, st, fr, nc in legs))
    return n_checked, where, frac, layout


def test_headline_launch_matches_oracle(gs, oracle, monkeypatch):
    """cfg2 as bench.py launches it: 10^6 chains x 1000 transitions of the README mixture, thin 100, default packing and
    slicing.  4 160 chain ids against the oracle."""
    per = 512                                                          # two chains per lane x 256 lanes

    def blocks(first, n_chunks):
        return [("first workgroups", 0, 640),
                ("an unsliced chunk in the middle of the first round", (first // 2) * per - 64, 640),
                ("across the whole-launch | sliced boundary", first * per - 320, 640),
                ("sliced chunks", (first + (n_chunks - first) // 2) * per - 100, 800),
                ("sliced chunks, far end", (n_chunks - 3) * per - 64, 640),
                ("ragged last chunk and its neighbour", 1_000_000 - 800, 800)]

    n_checked, where, frac, _ = _check_blocks(gs, oracle, "vmfmix_readme", 1_000_000, blocks, "screened_kernel<3, ScreenVmf<3, 3>>", per,
                                           (0.2, 0.5), monkeypatch, per_cu=5)
    assert n_checked >= 4000 and len(where) == 6


def test_curve_d50_launch_matches_oracle(gs, oracle, monkeypatch):
    """cfg4 d = 50 at its full size: 10^5 chains x 1000 transitions, every chunk of 64 chains sliced (8 hand-overs per chain)."""
    per = 64

    def blocks(first, n_chunks):
        return [("first chunks", 0, 160), ("middle", (n_chunks // 2) * per - 30, 160), ("ragged last chunk and its neighbours", 100_000 - 160, 160)]

    n_checked, where, frac, _ = _check_blocks(gs, oracle, "curve_d50", 100_000, blocks, "curvespec_kernel<4, 3, 10, +1>", per, (1.0, 1.0), monkeypatch, per_cu=3)
    assert n_checked == 480


@pytest.mark.parametrize("workload,kernel,per,m,per_cu", [("curve_d10", "curvespec_kernel<4, 1, 10", 64, 320, 3), ("curve_d24", "curvespec_kernel<4, 1, 10, +2>", 64, 200, 3),
                                                           ("curve_d200", "curvespec_kernel<16, 3, 10, +1>", 16, 48, 3)])
def test_curve_launches_match_oracle(gs, oracle, monkeypatch, workload, kernel, per, m, per_cu):
    """cfg4's other points at their full size (10^5 chains x 1000 transitions, every chunk sliced): three- and two-wavefront builds,
    four- and sixteen-lane groups, the packed segment evaluation with the full-curve copy of its loop."""

    def blocks(first, n_chunks):
        return [("first chunks", 0, m), ("middle", (n_chunks // 2) * per - per // 2, m), ("ragged last chunk and its neighbours", 100_000 - m, m)]

    n_checked, where, frac, _ = _check_blocks(gs, oracle, workload, 100_000, blocks, kernel, per, (1.0, 1.0), monkeypatch, per_cu=per_cu)
    assert n_checked == 3 * m


@pytest.mark.parametrize("workload,kernel,want_layout", [("bingham_d10", "screened_kernel<10, ScreenBinghamDiag<10>>", "chains"),
                                                          ("vmfmix_k10_kappa500", "screened_kernel<3, ScreenVmf<3, 10>>", "components")])
def test_one_chain_per_lane_launches_match_oracle(gs, oracle, monkeypatch, workload, kernel, want_layout):
    """cfg3 / cfg5 as bench.py launches them: 10^6 chains x 1000 transitions in the kernels' one-chain-per-lane BUILD (256-chain
    workgroups, no parked chain; round 4), the last workgroups sliced; cfg3 writes the reference's (chains, draws, dims) rows, held
    back in LDS until a run of them ends on a 32-byte sector."""
    per = 256

    def blocks(first, n_chunks):
        return [("first workgroups", 0, 384),
                ("an unsliced chunk in the middle", (first // 2) * per - 64, 384),
                ("across the whole-launch | sliced boundary", first * per - 192, 384),
                ("sliced chunks", (first + (n_chunks - first) // 2) * per - 100, 384),
                ("ragged last chunk and its neighbour", 1_000_000 - 400, 400)]

    n_checked, where, frac, layout = _check_blocks(gs, oracle, workload, 1_000_000, blocks, kernel, per, (0.005, 0.2), monkeypatch, per_cu=3)
    assert layout == want_layout and n_checked >= 1900 and len(where) == 5
