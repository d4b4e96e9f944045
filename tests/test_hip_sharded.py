"""Sharded ensembles equal the unsharded run BITWISE (the reference's scale-out is independent chains
whose results are concatenated in chain order, scripts/curve_vMF.py:205-267).

Two legs on the one GPU of the test box:
* in one process, the 8-way partition of BASELINE cfg5's node (ensemble.shard_bounds(n, r, 8), equal and ragged): eight
  samplers keyed by their chain offsets reproduce the single run's states and counters bit for bit;
* as separate PROCESSES: N gloo ranks share the GPU, each builds its shard with ensemble.sharded_sampler, and the gathered
  [d, n_total] of ensemble.gather_states / the totals of ensemble.reduce_sum equal the single-process run -- the whole
  multi-rank code path of a node job except the transport (gloo instead of RCCL, which refuses two ranks on one device).
  The box admits six processes on its GPU at once and this pytest process is one of them: N = 4 ranks, not 8; the
  world_size-8 collectives themselves run on CPU tensors in tests/test_ensemble_gloo.py.
"""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def gs():
    import torch
    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    import geosss_amd
    geosss_amd._lib.require_device()
    return geosss_amd


def _single_run(gs, n_total, n_steps):
    import bench
    pdf, d = bench.make_target(gs, "vmfmix_readme")
    x0 = gs.sample_sphere_device(d - 1, n_total, seed=0)
    s = gs.ShrinkageSphericalSliceSampler(pdf, x0.T, seed=3521, placement="packed")
    s.advance(n_steps)
    assert np.all(s.errors == 0)
    return pdf, s


@pytest.mark.parametrize("n_total", [32768, 32771])
def test_eight_way_partition_is_bitwise_the_single_run(gs, n_total):
    import torch
    from geosss_amd import ensemble
    pdf, full = _single_run(gs, n_total, 50)
    parts, rej, tries = [], 0, 0
    for r in range(8):
        lo, hi = ensemble.shard_bounds(n_total, r, 8)
        x0 = gs.sample_sphere_device(2, hi - lo, seed=0, chain_offset=lo)
        s = gs.ShrinkageSphericalSliceSampler(pdf, x0.T, seed=3521, chain_offset=lo, placement="packed")
        s.advance(50)
        parts.append(s.state_device)
        rej += s.n_reject
        tries += int(s._n_tries.sum().item())
        assert np.array_equal(s.n_reject_per_chain, full.n_reject_per_chain[lo:hi])
    assert torch.equal(torch.cat(parts, dim=1), full.state_device)
    assert rej == full.n_reject and tries == int(full._n_tries.sum().item())


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.parametrize("n_total", [32768, 32771])
def test_gloo_ranks_gather_the_single_run(gs, tmp_path, n_total):
    import torch
    world, n_steps = 4, 50
    out = str(tmp_path / "gathered.npz")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env["OMP_NUM_THREADS"] = "1"
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "tests", "sharded_rank.py"), str(n_total), str(n_steps), out]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300)   # (the launcher itself never touches the GPU)
    assert r.returncode == 0, r.stderr[-3000:]
    z = np.load(out)
    assert int(z["world"]) == world
    pdf, full = _single_run(gs, n_total, n_steps)
    assert np.array_equal(z["states"], full.state_device.cpu().numpy())             # bitwise, ragged shards included
    assert int(z["totals"][0]) == full.n_reject and int(z["totals"][1]) == int(full._n_tries.sum().item()) and int(z["totals"][2]) == 0
