"""Chain ensembles sharded over the GPUs of one node (one process per GPU).

Chains are independent (geosss/mcmc.py:382-401 has no exchange step; the reference's own
scale-out is independent processes, scripts/curve_vMF.py:205-267), so the data path needs no
collective: rank r owns the contiguous block of chain ids [r*n_local, (r+1)*n_local) and the
counter-based RNG stream makes every chain's numbers independent of the partition.  The only
communication is the final gather of states / reduction of statistics over RCCL (xGMI).
"""
import torch
import torch.distributed as dist


def world():
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def shard_bounds(n_total, rank=None, world_size=None):
    """Contiguous block [lo, hi) of chain ids owned by `rank`; blocks differ by at most one chain."""
    if rank is None or world_size is None:
        rank, world_size = world()
    base, rem = divmod(int(n_total), int(world_size))
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def gather_states(state_cm, group=None, always_collective=False, counts=None):
    """All-gather component-major states [d, n_local] -> [d, n_total] (rank order = chain order).

    `counts` = chains per rank, when the caller knows them (they are fixed for an ensemble: pass them and no
    size exchange or host synchronisation happens per call).  Equal shards of a small d gather each
    component row straight into its row of the result -- d all_gather_into_tensor calls (direct exchanges on
    the fully connected xGMI mesh), no transposes, no staging copies; large d gathers rows once and
    transposes once; ragged shards fall back to all_gather of padded blocks.
    A single rank returns its input unless always_collective (used to exercise the RCCL calls)."""
    rank, ws = world()
    if ws == 1 and not (always_collective and dist.is_available() and dist.is_initialized()):
        return state_cm
    d, n_local = state_cm.shape
    if counts is None:
        sizes = torch.tensor([n_local], dtype=torch.int64, device=state_cm.device)
        all_sizes = [torch.zeros_like(sizes) for _ in range(ws)]
        dist.all_gather(all_sizes, sizes, group=group)
        counts = [int(s.item()) for s in all_sizes]
    counts = [int(c) for c in counts]
    if len(counts) != ws or counts[rank] != n_local:
        raise ValueError("counts must list the chains of every rank")
    if len(set(counts)) == 1 and d <= 16:
        state_cm = state_cm.contiguous()
        out = torch.empty((d, ws * n_local), dtype=state_cm.dtype, device=state_cm.device)
        for j in range(d):
            dist.all_gather_into_tensor(out[j], state_cm[j], group=group)
        return out
    rows = state_cm.t().contiguous()  # [n_local, d]: concatenation over ranks is then contiguous
    if len(set(counts)) == 1:
        out = torch.empty((ws * n_local, d), dtype=rows.dtype, device=rows.device)
        dist.all_gather_into_tensor(out, rows, group=group)
    else:
        m = max(counts)
        pad = torch.zeros((m, d), dtype=rows.dtype, device=rows.device)
        pad[:n_local] = rows
        parts = [torch.empty_like(pad) for _ in range(ws)]
        dist.all_gather(parts, pad, group=group)
        out = torch.cat([p[:c] for p, c in zip(parts, counts)], dim=0)
    return out.t().contiguous()


def reduce_sum(t, group=None):
    """Sum a small statistics tensor over ranks (n_reject totals, mode counts, moments)."""
    _, ws = world()
    if ws > 1:
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    return t


def sharded_sampler(cls, distribution, n_total, seed, d=None, init_seed=0, **kwargs):
    """Build this rank's sampler for an ensemble of n_total chains with uniform initial states:
    chain ids and initial states are those of the unsharded ensemble restricted to the shard."""
    from .sphere import sample_sphere_device
    lo, hi = shard_bounds(n_total)
    d = d or distribution.d
    x0 = sample_sphere_device(d - 1, hi - lo, seed=init_seed, chain_offset=lo, device=kwargs.get("device"))
    return cls(distribution, x0.t(), seed, chain_offset=lo, **kwargs)
