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


def trazar_sharded(selected_func, scenario, show, step, divisor, user_choice, *, thetas=None, starts=None, record=None,
                   dtype=0, dst=0, group=None, device=None, **kw):
    """rt_bench.trazar (RT_bench.py:766-948) with the rays of ONE call split over the ranks of a torch.distributed group --
    one process per GPU, the reference's outer loop over rays (:807) cut into `world` interleaved shards (rank r traces rays
    r, r + world, ...: every rank gets the same mix of short and long rays).  Each rank builds the field on its own device
    (genZ + interpolacion, a few ms) and runs its shard; there is no collective on the data path.  The results -- d_ray,
    compute_times, errors and, when `record` is set ("full" or a stride), the recorded rows -- are gathered to `dst` in the
    caller's ray order (over RCCL when the group's backend is nccl, device to device).  Returns what trazar returns on `dst`,
    None on the other ranks.  Rays are independent, so the gathered arrays are bit-identical to an unsharded call.

    thetas / starts ((R, 2) or (2,)) replace the preset of `user_choice` like in trazar; the interface scenario's exit-angle
    errors need record="full"."""
    import os
    import torch
    import torch.distributed as dist
    from . import _lib
    from . import rt_bench as rb
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    on_gpu = dist.get_backend(group) == "nccl"
    if device is None:
        device = int(os.environ.get("LOCAL_RANK", rank))
    _lib.check(_lib.lib().rtmi_set_device(int(device)))
    c = rb.constants(user_choice)
    theta_all = np.asarray(c[2][:c[1]] if thetas is None else thetas, dtype=np.float64)
    R_total = len(theta_all)
    th = np.ascontiguousarray(theta_all[rank::world])
    st = None
    if starts is not None:
        st = np.asarray(starts, dtype=np.float64)
        st = st if st.ndim == 1 else np.ascontiguousarray(st[rank::world])
    fld = rb.Field.build(scenario, c[5:9], rb.DELTA, dtype)
    try:
        z, grd = rb.FieldSpline(fld, "n"), (rb.FieldSpline(fld, "dy"), rb.FieldSpline(fld, "dx"))
        s_ray, d_ray, ctimes, errors = rb.trazar(selected_func, z, grd, False, step, divisor, user_choice, thetas=th, starts=st,
                                                 record=record, **kw)
    finally:
        fld.close()
    loc = np.concatenate([d_ray, ctimes[None], errors[None]] + ([s_ray.reshape(-1, len(th))] if s_ray is not None else []), axis=0)
    Rmax = (R_total + world - 1) // world
    t = torch.from_numpy(np.ascontiguousarray(loc))
    if t.shape[1] < Rmax:                                            # ragged split: pad the short ranks
        t = torch.cat((t, torch.full((t.shape[0], Rmax - t.shape[1]), float("nan"), dtype=t.dtype)), 1)
    if on_gpu:
        t = t.to(torch.device("cuda", int(device)))
    out = [torch.empty_like(t) for _ in range(world)] if rank == dst else None
    dist.gather(t.contiguous(), out, dst=dst, group=group)
    if rank != dst:
        return None
    allr = interleave(out, R_total).cpu().numpy()                    # [5 (+ rows*6), R_total] in ray order
    d_all, ct_all, err_all = allr[:3], allr[3], allr[4]
    s_all = allr[5:].reshape(s_ray.shape[0], 6, R_total) if s_ray is not None else None
    if show and s_all is not None and c[9]:
        angsim, angreal = rb.snell_angles(s_all, d_all, theta_all)
        for k in range(R_total):
            i = int(d_all[2, k])
            f = rb._format_num
            print(f"Coords: [ {f(s_all[i, 0, k])} , {f(s_all[i, 1, k])} ] | SimAng: {f(angsim[k])} | "
                  f"SnellAng: {f(angreal[k])} | Err: {f(err_all[k])} | InitAng: {f(theta_all[k] * 180 / np.pi)}")
    return s_all, d_all, ct_all, err_all
