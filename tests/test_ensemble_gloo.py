"""Multi-process coverage of the sharding layer on CPU (gloo, world_size 2 and 3): block bounds,
the final gather (equal and ragged shards) and the statistics reduction.  The GPU data path has no
collective; partition invariance of the chains themselves is covered by the gpu tests."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, n_total, d, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from geosss_amd import ensemble
        lo, hi = ensemble.shard_bounds(n_total)
        full = torch.arange(n_total * d, dtype=torch.float64).reshape(n_total, d)  # row c = chain c
        local_cm = full[lo:hi].t().contiguous()                                     # [d, n_local]
        got = ensemble.gather_states(local_cm)
        assert got.shape == (d, n_total)
        assert torch.equal(got, full.t().contiguous())
        stat = torch.tensor([hi - lo, rank], dtype=torch.int64)
        ensemble.reduce_sum(stat)
        assert int(stat[0]) == n_total and int(stat[1]) == world * (world - 1) // 2
        np.save(os.path.join(out_dir, f"ok_{rank}.npy"), np.array([lo, hi]))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,n_total", [(2, 1000), (2, 1001), (3, 64), (8, 32768), (8, 32771)])  # (8: the node of BASELINE cfg5)
def test_gather_and_bounds(tmp_path, world, n_total):
    port = _free_port()
    mp.spawn(_worker, args=(world, port, n_total, 3, str(tmp_path)), nprocs=world, join=True)
    bounds = [np.load(tmp_path / f"ok_{r}.npy") for r in range(world)]
    assert bounds[0][0] == 0 and bounds[-1][1] == n_total
    for a, b in zip(bounds, bounds[1:]):
        assert a[1] == b[0]
    sizes = [b[1] - b[0] for b in bounds]
    assert max(sizes) - min(sizes) <= 1


def test_shard_bounds_single_process():
    from geosss_amd.ensemble import shard_bounds, world
    assert world() == (0, 1)
    assert shard_bounds(10) == (0, 10)
    parts = [shard_bounds(10, r, 4) for r in range(4)]
    assert parts == [(0, 3), (3, 6), (6, 8), (8, 10)]
