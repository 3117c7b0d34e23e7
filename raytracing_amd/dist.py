"""Ray sharding across GPUs: rays are independent (the reference's outer loop,
RT_bench.py:807, carries nothing between rays), so any partition gives the same bits: contiguous blocks
(shard_range / fan_shard / gather_blocks) or the interleaved split bench.py uses for load balance (interleave);
the field is rebuilt per rank (<= 26 MB), and the only collective is the read-back gather.
One process per GPU; torch.distributed is plumbing ("nccl" == RCCL over xGMI on ROCm, "gloo" on CPU)."""
import numpy as np


def shard_range(R, rank, world):
    """Contiguous block [lo, hi) of rank `rank`; blocks differ by at most one ray and cover [0, R)."""
    base, rem = divmod(int(R), int(world))
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def fan_shard(theta_lo, theta_hi, R_total, rank, world):
    """Launch angles of this rank's block of linspace(theta_lo, theta_hi, R_total), bit-identical to slicing
    the full array (numpy.linspace: arange(n)*step + start, last element = stop)."""
    lo, hi = shard_range(R_total, rank, world)
    step = (theta_hi - theta_lo) / (R_total - 1) if R_total > 1 else 0.0
    th = np.arange(lo, hi, dtype=np.float64) * step + theta_lo
    if hi == R_total and R_total > 1:
        th[-1] = theta_hi
    return th


def gather_blocks(local, R_total, dst=0, group=None):
    """Gather per-rank blocks [..., R_local] of a torch tensor to `dst` in ray order -> [..., R_total] (None
    elsewhere).  Blocks may differ in length by one, so they are padded to the longest for the collective."""
    import torch
    import torch.distributed as dist
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    longest = shard_range(R_total, 0, world)[1]
    pad = longest - local.shape[-1]
    buf = torch.nn.functional.pad(local, (0, pad)) if pad else local.contiguous()
    out = [torch.empty_like(buf) for _ in range(world)] if rank == dst else None
    dist.gather(buf, out, dst=dst, group=group)
    if rank != dst:
        return None
    parts = []
    for r in range(world):
        lo, hi = shard_range(R_total, r, world)
        parts.append(out[r][..., :hi - lo])
    return torch.cat(parts, dim=-1)


def interleave(per_rank, R_total):
    """Inverse of the interleaved partition bench.py uses (rank r owns rays r, r+world, r+2*world, ...): per_rank is
    the gathered list of [..., Rmax] tensors, Rmax = ceil(R_total / world), short ranks padded at the end.
    Returns [..., R_total] in ray order."""
    import torch
    world = len(per_rank)
    stacked = torch.stack(list(per_rank), dim=-1)                 # [..., Rmax, world]: slot k, rank r -> ray k*world + r
    flat = stacked.reshape(*stacked.shape[:-2], stacked.shape[-2] * world)
    return flat[..., :R_total]


def max_over_ranks(seconds, device=None, group=None):
    import torch
    import torch.distributed as dist
    t = torch.tensor([seconds], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
    return float(t.item())


def sum_over_ranks(value, device=None, group=None):
    import torch
    import torch.distributed as dist
    t = torch.tensor([int(value)], dtype=torch.int64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    return int(t.item())
