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


def agree_ok(ok, device=None, group=None):
    """True on every rank iff `ok` is true on every rank (one all_reduce): called before a collective that a failed rank
    would never enter, so that the others raise instead of waiting for it forever."""
    import torch
    import torch.distributed as dist
    t = torch.tensor([0 if ok else 1], dtype=torch.int32, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
    return int(t.item()) == 0


def _to_callers_order(t, perm):
    """Slot order -> the caller's ray order along the last axis (sort_rays: slot k holds the caller's ray perm[k])."""
    if perm is None:
        return t
    import torch
    out = torch.empty_like(t)
    out[..., perm.long()] = t
    return out


def trazar_sharded(selected_func, scenario, show, step, divisor, user_choice, *, thetas=None, starts=None, record=None,
                   dtype=0, dst=0, group=None, device=None, **kw):
    """rt_bench.trazar (RT_bench.py:766-948) with the rays of ONE call split over the ranks of a torch.distributed group --
    one process per GPU, the reference's outer loop over rays (:807) cut into `world` interleaved shards (rank r traces rays
    r, r + world, ...: every rank gets the same mix of short and long rays).  Each rank builds the field on its own device
    (genZ + interpolacion, a few ms) and runs its shard; there is no collective on the data path.  The results -- d_ray,
    compute_times, errors and, when `record` is set ("full" or a stride), the recorded rows -- are gathered to `dst` in the
    caller's ray order.  With the nccl backend (RCCL over xGMI) the gather is device to device: the payload is assembled from
    zero-copy views of the batch's HBM arrays (Batch.device_tensors(): the SoA state and the record `s_ray[rows][6][R_local]`),
    nothing of the trajectory passes through the host before it reaches `dst`.  With gloo (CPU rehearsal) the same payload is
    read back to the host first.  Returns what trazar returns on `dst`, None on the other ranks.  Rays are independent, so
    the gathered arrays are bit-identical to an unsharded call.

    A rank whose local trace fails (RtmiError, allocation) does not leave the others blocked in the gather: the ranks agree on
    success with one all_reduce first and all raise.  A rank with no rays (world > number of rays) contributes padding.

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
    cdev = torch.device("cuda", int(device)) if on_gpu else torch.device("cpu")
    pl = rb.trazar_plan(user_choice, step, divisor, thetas, starts, kw.get("box"), kw.get("gamma"), kw.get("max_size"), record)
    theta_all, R_total = pl["theta_v"], pl["ray_count"]
    th = np.ascontiguousarray(theta_all[rank::world])
    R_local = len(th)
    Rmax = (R_total + world - 1) // world
    st = None
    if starts is not None:
        st = np.asarray(starts, dtype=np.float64)
        st = st if st.ndim == 1 else np.ascontiguousarray(st[rank::world])
    rows = pl["rec_rows"]
    fdt = torch.float64 if dtype == rb.F64 else torch.float32
    head = rows_t = None          # [5, R_local] fp64: d_ray (3), compute_times, errors; [rows*6, R_local] of the batch's dtype
    fld = b = None
    err = None
    try:
        if R_local > 0:
            _lib.check(_lib.lib().rtmi_set_device(int(device)))
            fld = rb.Field.build(scenario, rb.constants(user_choice)[5:9], rb.DELTA, dtype)
            z, grd = rb.FieldSpline(fld, "n"), (rb.FieldSpline(fld, "dy"), rb.FieldSpline(fld, "dx"))
            _, d_ray, ctimes, errors, b = rb.trazar(selected_func, z, grd, False, step, divisor, user_choice, thetas=th, starts=st,
                                                    record=record, return_batch=True, read_rows=not on_gpu, **kw)
            if on_gpu:
                t = b.device_tensors()
                perm = t.get("perm")
                d_dev = torch.stack((t["dist_real"], t["dist_sim"], t["istep"].to(torch.float64)))          # (:888-890)
                small = torch.from_numpy(np.stack((ctimes, errors))).to(cdev)                                # 16 bytes per ray
                head = torch.cat((_to_callers_order(d_dev, perm), small), 0)
                if rows:
                    rows_t = _to_callers_order(t["s_ray"].reshape(rows * 6, R_local), perm)                # a view unless sort_rays
            else:
                head = torch.from_numpy(np.concatenate([d_ray, ctimes[None], errors[None]], axis=0))
                if rows:
                    rows_t = torch.from_numpy(b.rows().reshape(rows * 6, R_local)).to(fdt)
        else:
            head = torch.empty((5, 0), dtype=torch.float64, device=cdev)
            rows_t = torch.empty((rows * 6, 0), dtype=fdt, device=cdev) if rows else None
    except Exception as e:      # agreed on below: no rank enters the gather unless all can
        err = e
    try:
        if not agree_ok(err is None, cdev, group):
            raise RuntimeError(f"trazar_sharded: the local trace failed on {'this rank' if err is not None else 'another rank'}"
                               f" (rank {rank} of {world})") from err

        def gather_padded(t):
            if t.shape[1] < Rmax:                                            # ragged split: pad the short ranks
                t = torch.cat((t, torch.full((t.shape[0], Rmax - t.shape[1]), float("nan"), dtype=t.dtype, device=t.device)), 1)
            out = [torch.empty_like(t) for _ in range(world)] if rank == dst else None
            dist.gather(t.contiguous(), out, dst=dst, group=group)
            return interleave(out, R_total) if rank == dst else None         # [.., R_total] in ray order
        head_all = gather_padded(head)
        rows_all = gather_padded(rows_t) if rows else None
        if on_gpu:
            torch.cuda.synchronize(cdev)                                     # the batch's memory is released below
    finally:
        if b is not None:
            b.close()
        if fld is not None:
            fld.close()
    if rank != dst:
        return None
    head_np = head_all.cpu().numpy()
    d_all, ct_all, err_all = head_np[:3], head_np[3], head_np[4]
    s_all = rows_all.to(torch.float64).cpu().numpy().reshape(rows, 6, R_total) if rows else None
    if show and s_all is not None and pl["op_interface"]:
        rb.print_exit_table(s_all, d_all, err_all, theta_all)
    return s_all, d_all, ct_all, err_all
