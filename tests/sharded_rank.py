"""One rank of the sharded-ensemble rehearsal (tests/test_hip_sharded.py starts N of these under torch.distributed.run, gloo
backend, all on the box's one GPU): build this rank's shard with ensemble.sharded_sampler, advance it, gather the states and
reduce the rejection count the way a multi-GPU job does, and let rank 0 write what it gathered."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    n_total, n_steps, out_path = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
    import torch
    import torch.distributed as dist
    import bench
    import geosss_amd as gs
    from geosss_amd import ensemble
    torch.cuda.set_device(0)                       # every rank on the same card: this is a rehearsal of the code path, not of xGMI
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    pdf, d = bench.make_target(gs, "vmfmix_readme")
    s = ensemble.sharded_sampler(gs.ShrinkageSphericalSliceSampler, pdf, n_total, seed=3521, placement="packed")
    lo, hi = ensemble.shard_bounds(n_total)
    assert s.n_chains == hi - lo and s.chain_offset == lo
    s.advance(n_steps)
    final = ensemble.gather_states(s.state_device)                      # [d, n_total] on every rank
    totals = torch.stack([s._n_reject.sum(), s._n_tries.sum(), (s._err != 0).sum().to(torch.int64)])
    ensemble.reduce_sum(totals)
    assert final.shape == (d, n_total)
    mine = final[:, lo:hi]
    assert torch.equal(mine, s.state_device)                             # a rank finds its own block where its chain ids say
    if rank == 0:
        np.savez(out_path, states=final.cpu().numpy(), totals=totals.cpu().numpy(), world=world)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
