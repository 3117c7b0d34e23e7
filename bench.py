#!/usr/bin/env python3
"""bench.py -- ray-steps/sec of the propagation hot path (BASELINE.json metric) on N MI355X GPUs.

A "step" is one full propagation (initial conditions + every DELTA_S step to termination) of one batch of
synthetic rays per GPU.  N=1 workload: vert_heterogeneous, 1 048 576 rays, op6 (HySA), default DELTA_S,
fp64 -- the configuration the metric is quoted on (SURVEY.md 8d "north-star run").  For N>1 every rank
owns 1 048 576 rays of an N-times finer fan, interleaved (ray k*N + rank), so per-GPU work is fixed
("weak") and balanced; there is no data-path collective (rays are independent).

  python bench.py --gpus 1 --steps 5 --warmup 1
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
      bench.py --gpus N --steps K --warmup W

Rank 0 prints ONE JSON line.  `roofline` prices the advance kernel at SURVEY.md 8d's algorithmic bytes per
ray-step against 8 TB/s; `cpu_baseline` times the CPU oracle (oracle/, a port of the reference's algorithm)
on this host's cores over a bounded sample of the same workload.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0   # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
FP64_VECTOR_PEAK_TFLOPS = 78.6   # MI355X vector fp64 (half the 157.3 TF fp32 vector peak of MI355X_MICROARCH.md)
# fp64 flops per ray-step, from rocprofv3 SQ_INSTS_VALU_{FMA,MUL,ADD}_F64 per wave-step of the op6 kernel
# (profiles/r01_d_final_pmc_summary.txt: 100 fma + 68 mul + 24 add wave-instructions per 64-ray wave-step)
FLOPS_PER_RAY_STEP = {(6, "f64"): 2 * 100 + 68 + 24}
SCEN = {"vert_heterogeneous": dict(choice="3", theta=(0.0, np.pi / 2), start=(-2.0, -2.0), gamma=1),
        "anisotropy": dict(choice="4", theta=(0.0, np.pi / 2), start=(-2.0, -2.0), gamma=3),
        "interface": dict(choice="1", theta=(2 * np.pi / 60, np.pi / 2), start=(-2.0, -2.0), gamma=1),
        "fisheye": dict(choice="2", theta=(np.pi / 4, 3 * np.pi / 4), start=(1.0, 0.0), gamma=1)}


def alg_bytes_per_step(dtype, stride):
    """9 state values in + 9 out + 36 gathered coefficients, + 7 recorded values per stored row."""
    e = 8 if dtype == "f64" else 4
    return (9 + 9 + 36) * e + (7 * e / stride if stride else 0.0)


def fan(scen, R_total, rank, world):
    lo, hi = SCEN[scen]["theta"]
    step = (hi - lo) / (R_total - 1)
    idx = np.arange(rank, R_total, world, dtype=np.float64)
    th = idx * step + lo
    if (R_total - 1) % world == rank:
        th[-1] = hi
    return th


def cpu_baseline(args, rb, budget_s):
    """The oracle (kind "port") on this host's cores, same scenario/method/DELTA_S, a subsample of the fan."""
    from oracle import rt_oracle as O
    sc = SCEN[args.scenario]
    lim = rb.constants(sc["choice"])[5:9]
    fld = O.Field("vert_heterogeneous" if args.scenario == "anisotropy" else args.scenario, lim, rb.DELTA)
    cores = min(O.max_threads(), os.cpu_count() or 1)
    step, max_size = trace_step(args, rb)
    R = 64 * cores
    rate, used = 0.0, 0.0
    while True:
        th = np.linspace(*sc["theta"], R)
        t0 = time.perf_counter()
        r = O.trazar(fld, args.method, sc["gamma"], step, max_size, lim, sc["start"][0], sc["start"][1], th,
                     record_stride=0, nthreads=cores)
        dt = time.perf_counter() - t0
        used += dt
        rate = r["steps"] / dt
        if dt >= 0.4 * budget_s or used >= budget_s or R >= args.rays:
            break
        R = int(min(args.rays, max(2 * R, R * 0.6 * budget_s / max(dt, 1e-3))))
    return {"value": rate, "unit": "ray-steps/s", "cores": cores, "kind": "port",
            "sample": f"{args.scenario} op{args.method} fp64, {R} rays of the same fan, {r['steps']} ray-steps in "
                      f"{dt:.2f} s, OpenMP over rays, final-state mode"}


def trace_step(args, rb):
    c = rb.constants(SCEN[args.scenario]["choice"])
    if args.scenario == "fisheye":
        return 2 * np.pi / 303, rb.N * 304          # calibrated op6 step (RT_bench.py:1443, :1450)
    return rb.DELTA_S, int(np.ceil(c[4] / rb.DELTA_S) + 1)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--rays", type=int, default=1048576, help="rays per GPU")
    ap.add_argument("--scenario", default="vert_heterogeneous", choices=sorted(SCEN))
    ap.add_argument("--method", type=int, default=None)
    ap.add_argument("--dtype", default="f64", choices=["f64", "f32"])
    ap.add_argument("--record", default="full", help="full (the reference's s_ray layout, default) | none | stride:N")
    ap.add_argument("--mode", default="lane", choices=["lane", "refill"],
                    help="lane: one lane per ray; refill: persistent waves with ballot/prefix lane refill")
    ap.add_argument("--order", default="fan", choices=["fan", "shuffled"],
                    help="ray order inside the batch: the sorted fan, or a seeded random permutation of it")
    ap.add_argument("--refill-min", type=int, default=0)
    ap.add_argument("--sort", action="store_true", help="sort rays inside the batch by launch cell and angle (sort_rays)")
    ap.add_argument("--field-path", default="auto", choices=["auto", "lds", "global"],
                    help="auto (default): LDS tile when recording densely, else global; lds / global force one")
    ap.add_argument("--rec-rows", type=int, default=0, help="rows to allocate (0 = from max_size; full: 3072 for vert)")
    ap.add_argument("--block", type=int, default=0)
    ap.add_argument("--chunk", type=int, default=0,
                    help="0: one launch runs every ray to termination (default); N: the host advances the batch N "
                         "DELTA_S steps per launch (state round-trips HBM between launches), N=1 is one step at a time")
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="CPU baseline budget (0 = skip)")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="torch.distributed backend for N>1 (nccl = RCCL over xGMI; gloo only to rehearse the "
                         "multi-rank path on a box with fewer GPUs than ranks)")
    ap.add_argument("--all-on-device", type=int, default=None,
                    help="rehearsal only: every rank uses this HIP device instead of LOCAL_RANK")
    args = ap.parse_args()
    if args.method is None:
        args.method = 11 if args.scenario == "anisotropy" else 6

    import torch
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        sys.exit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}; launch with torch.distributed.run")
    if not torch.cuda.is_available():
        sys.exit("bench.py: no HIP device (raytracing_amd has no CPU path)")
    if args.all_on_device is not None:
        local = args.all_on_device
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    cdev = dev if args.backend == "nccl" else torch.device("cpu")     # where the collectives' tensors live
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group("gloo")

    from raytracing_amd import rt_bench as rb
    from raytracing_amd import _lib
    from raytracing_amd import dist as rd
    _lib.check(_lib.lib().rtmi_set_device(local))

    sc = SCEN[args.scenario]
    lim = rb.constants(sc["choice"])[5:9]
    dtype = rb.F64 if args.dtype == "f64" else rb.F32
    stride = 0 if args.record == "none" else (1 if args.record == "full" else int(args.record.split(":")[1]))
    step, max_size = trace_step(args, rb)
    rec_rows = args.rec_rows
    if stride and not rec_rows and args.scenario in ("vert_heterogeneous", "anisotropy"):
        rec_rows = (3072 + stride - 1) // stride     # the fan's longest ray takes 2 938 steps
    R_total = args.rays * world
    th = fan(args.scenario, R_total, rank, world)
    if args.order == "shuffled":
        th = np.random.default_rng(1234 + rank).permutation(th)
    fld = rb.Field.build(args.scenario, lim, rb.DELTA, dtype)
    def make_batch(stride_, rec_rows_):
        return rb.Batch(fld, args.method, step, max_size, lim, sc["gamma"], th, sc["start"][0], sc["start"][1],
                        record_stride=stride_, rec_rows=rec_rows_, block_size=args.block,
                        launch_mode=1 if args.mode == "refill" else 0, refill_min=args.refill_min,
                        field_path={"auto": 0, "global": 1, "lds": 2}[args.field_path], sort_rays=args.sort)

    try:
        batch = make_batch(stride, rec_rows)
    except _lib.RtmiError as e:
        if stride != 1:
            raise
        # the full trajectory (176 GB at 1 M rays) did not fit this device: keep every 16th row instead
        print(f"bench.py: full trajectory record failed ({e}); falling back to record=stride:16", file=sys.stderr)
        args.record, stride = "stride:16", 16
        rec_rows = (rec_rows + 15) // 16 if rec_rows else 0
        batch = make_batch(stride, rec_rows)

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    longest = {"vert_heterogeneous": 2953, "anisotropy": 2900}.get(args.scenario, max_size - 1)

    def one_pass():
        batch.reset()
        if args.chunk <= 0:
            batch.run()
        else:   # enough launches for the longest ray of the fan; launches after the last ray stopped exit at once
            for _ in range((longest + args.chunk - 1) // args.chunk + 1):
                batch.step(args.chunk)
            batch.sync()

    for _ in range(args.warmup):
        one_pass()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        one_pass()
    barrier()
    dt = time.perf_counter() - t0
    st = batch.stats()                       # counters and kernel time of the LAST pass (reset clears them)
    steps_per_pass = st["ray_steps"]
    kern_ms = st["kernel_ms"] / max(st["launches"], 1)
    if world > 1:
        dt = rd.max_over_ranks(dt, cdev)
        total_steps = rd.sum_over_ranks(steps_per_pass, cdev) * args.steps
        # read-back of the sharded layout: d_ray gathered to rank 0 over RCCL (outside the timed region)
        d_local = torch.as_tensor(batch.d_ray(), device=cdev)
        g = [torch.empty_like(d_local) for _ in range(world)] if rank == 0 else None
        dist.gather(d_local, g, dst=0)
        if rank == 0:
            d_all = torch.stack(g, dim=-1).reshape(3, R_total)      # ray k*world + r  <-  rank r, slot k
            assert int(d_all[2].sum().item()) * args.steps == total_steps
        if args.backend == "nccl":
            # end points gathered device-to-device: zero-copy views of the library's SoA state into RCCL
            try:
                dt_ = batch.device_tensors()
                xy = torch.stack((dt_["x"], dt_["y"]))                 # [2, R] on this rank's GPU
                gx_ = [torch.empty_like(xy) for _ in range(world)] if rank == 0 else None
                dist.gather(xy, gx_, dst=0)
                if rank == 0:
                    xy_all = torch.stack(gx_, dim=-1).reshape(2, R_total)
                    assert torch.isfinite(xy_all).all()
            except Exception as e:   # the timed result is already in hand; report and carry on
                print(f"bench.py: device-side gather skipped: {e}", file=sys.stderr)
    else:
        total_steps = steps_per_pass * args.steps

    if rank == 0:
        balg = alg_bytes_per_step(args.dtype, stride)
        achieved = balg * steps_per_pass / (kern_ms * 1e-3) / 1e9
        out = {
            "metric": "ray-steps/sec (whole node) on vert_heterogeneous, 1M rays; % HBM roofline",
            "value": total_steps / dt, "unit": "ray-steps/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": 1e3 * dt / args.steps, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": f"{args.scenario}, {args.rays} rays/GPU (fan linspace over {R_total} rays, "
                                   f"interleaved across ranks), op{args.method}, DELTA_S={step:.12g}, "
                                   f"box={tuple(float(v) for v in lim)}, record={args.record}",
                       "rays_per_gpu": args.rays, "ray_steps_per_pass_rank0": int(steps_per_pass),
                       "method": f"op{args.method}", "record": args.record, "launch_mode": args.mode,
                       "ray_order": args.order, "sort_rays": bool(args.sort), "field_path": args.field_path, "steps_per_launch": args.chunk or "all", "parallelism": f"ray-shard x{world}"},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": None,
                         "kernel": "k_trace_refill" if args.mode == "refill" else "k_advance", "kernel_ms": kern_ms, "alg_bytes_per_ray_step": balg,
                         "ray_steps_per_launch": int(steps_per_pass), "vgprs": st["vgprs"]},
        }
        fl = FLOPS_PER_RAY_STEP.get((args.method, args.dtype))
        if fl:   # what actually limits the kernel: the fp64 vector ALU (reported beside the contract's HBM line)
            tf = fl * steps_per_pass / (kern_ms * 1e-3) / 1e12
            out["roofline"]["valu_fp64"] = {"achieved": tf, "peak": FP64_VECTOR_PEAK_TFLOPS, "unit": "TFLOP/s",
                                            "frac": tf / FP64_VECTOR_PEAK_TFLOPS, "flops_per_ray_step": fl}
        prof = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(prof):
            try:
                tr = json.load(open(prof))
                key = f"{args.scenario}:{args.rays}:{args.record}:{args.dtype}:op{args.method}"
                if key in tr:
                    out["roofline"]["traffic"] = tr[key]
            except Exception:
                pass
        if world == 1 and args.cpu_seconds > 0:
            out["cpu_baseline"] = cpu_baseline(args, rb, args.cpu_seconds)
        print(json.dumps(out), flush=True)
    batch.close()
    fld.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
