#!/usr/bin/env python3
"""bench.py -- ray-steps/sec of the propagation hot path (BASELINE.json metric) on N MI355X GPUs.

A "step" is one full propagation (initial conditions + every DELTA_S step to termination) of one batch of
synthetic rays per GPU.  N=1 workload: vert_heterogeneous, 1 048 576 rays, op6 (HySA), default DELTA_S,
fp64, full trajectory record -- the configuration the metric is quoted on (SURVEY.md 8d "north-star run").

Multi-GPU (one process per GPU, no data-path collective -- rays are independent):
  strong (default; --total-rays R, R = 1 048 576 unless given): the SAME R-ray fan at every N -- the north star's
                    "vert_heterogeneous, 1M rays" at 1/2/4/8 GPUs -- rank r owns rays r, r+N, r+2N, ...;
                    `value`, `ms_per_step` and `scaling: "strong"` of the JSON line are this run's.  For N > 1 the same ranks
                    then time the weak configuration too (1 048 576 rays per GPU) and report it as the secondary record
                    `weak` of the same line (--no-weak skips it);
  weak   (--rays R): every rank owns R rays of an N-times finer fan, interleaved (ray k*N + rank), so per-GPU work is
                    fixed and balanced; `scaling: "weak"`.

  python bench.py --gpus 1 --steps 5 --warmup 1
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
      bench.py --gpus N --steps K --warmup W [--total-rays 1048576]

Single-GPU rehearsals of the N-GPU run (no torch.distributed involved):
  --emulate-world N --emulate-rank r   run rank r's shard of the N-way split on this GPU: the per-GPU operating point of an
                    N-GPU run (there is no collective in the timed region, so the per-shard rate IS the scaling curve)
  --force-dist      initialise torch.distributed (--backend nccl = RCCL) with world_size 1 and run the all_reduce / gather
                    of the N>1 path through it: the part of the multi-GPU path one MI355X can execute

Rank 0 prints ONE JSON line.
  parity_check  after the timed passes: every 512th ray of the batch that was just timed (d_ray, final state and -- with a
                full record -- every recorded row, read from the device copy) against the CPU oracle on the same launch
                angles; the run exits non-zero if a step count differs or an fp64 value is further than 1e-9 (relative)
                from the oracle's.
  roofline      what bounds the advance kernel, as a fraction <= 1 of a hardware peak: HBM bytes per launch (rocprofv3
                counters, profiles/traffic.json; or, without a profile of this configuration, the bytes the launch must
                write and read: recorded rows + state) over the kernel time measured here with HIP events, against
                8 TB/s -- and the vector ALU's issue time from the profiled instruction counts (4 cycles per fp64
                wave-instruction, 2 per other) against 1024 SIMDs at 2.4 GHz.  `bound` names the larger.  SURVEY.md 8d's ALGORITHMIC byte model (state in/out + coefficient gather per ray-step) is kept
                under `alg_model`; it is not a bound for this kernel (state lives in registers across all steps of a
                ray, the gather is served from LDS/L2), which is why it exceeds the HBM peak.
  cpu_baseline  the CPU oracle (oracle/, a port of the reference's algorithm) on this host's cores over a bounded
                sample of the same workload.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0            # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured for a float4 copy)
HBM_FILL_GBS = 6730.0            # profiles/r02_f_hbm_write_ceiling.txt: a bare fill kernel writing the record's rows in k_advance's pattern
SIMDS = 256 * 4                  # CUs x SIMDs
PEAK_CLOCK_HZ = 2.4e9            # MI355X_MICROARCH.md: max clock
SCEN = {"vert_heterogeneous": dict(choice="3", theta=(0.0, np.pi / 2), start=(-2.0, -2.0), gamma=1),
        "anisotropy": dict(choice="4", theta=(0.0, np.pi / 2), start=(-2.0, -2.0), gamma=3),
        "interface": dict(choice="1", theta=(2 * np.pi / 60, np.pi / 2), start=(-2.0, -2.0), gamma=1),
        "fisheye": dict(choice="2", theta=(np.pi / 4, 3 * np.pi / 4), start=(1.0, 0.0), gamma=1)}


def alg_bytes_per_step(dtype, stride):
    """9 state values in + 9 out + 36 gathered coefficients, + 7 recorded values per stored row."""
    e = 8 if dtype == "f64" else 4
    return (9 + 9 + 36) * e + (7 * e / stride if stride else 0.0)   # SURVEY counts n_ray among the recorded values


def min_hbm_bytes(dtype, stride, ray_steps, rays, method, n_ray):
    """Bytes one launch cannot avoid moving: the recorded rows (6 values per stored row, 7 with n_ray) and the ray state
    once in and once out (6 fp64 accumulators + 3 values, +4 for op7, + istep + alive).  The field (<= 26 MB) is
    L2/MALL-resident."""
    e = 8 if dtype == "f64" else 4
    rows = ray_steps / stride if stride else 0.0
    return rows * (7 if n_ray else 6) * e + 2.0 * rays * (48 + (7 if method == 7 else 3) * e + 5)


# quantities that share a scale, by the length of an array's quantity axis (the second to last: [.., quantity, ray]):
#   9 final state  x y | theta | n | dn/dx dn/dy | p_x p_y | T        6 recorded row  x y | p_x p_y | T | theta
#   3 d_ray        dist_real dist_sim | last row                     2 d_ray[:2]
QUANTITY_GROUPS = {9: ((0, 1), (2,), (3,), (4, 5), (6, 7), (8,)), 6: ((0, 1), (2, 3), (4,), (5,)), 3: ((0, 1), (2,)), 2: ((0, 1),)}


def parity_relerr(a, b):
    """max |a - b| / scale with a scale PER QUANTITY: the largest |b| of that quantity (vector quantities -- position, momentum,
    gradient -- as one; the gradient's at least the index itself) over the whole fixture.  "1e-9 relative" then means 1e-9 of p_x ~ 0.05 or T ~ 0.4 as much as of
    x ~ 5; dividing by max(|b|, 1) instead would hold everything below 1 in magnitude to an ABSOLUTE 1e-9."""
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    if a.size == 0:
        return 0.0
    groups = QUANTITY_GROUPS[a.shape[-2]]
    worst = 0.0
    for g in groups:
        sel = list(g)
        bb, aa = np.take(b, sel, axis=-2), np.take(a, sel, axis=-2)
        scale = float(np.max(np.abs(bb)))
        if a.shape[-2] == 9 and g == (4, 5):
            # grad n at the end points: where every ray of the sample ends in a constant part of the medium the gradient there is
            # the tail of a global spline (1e-20 .. 1e-113), no scale to be relative to; its natural one is the index per unit
            # length (what it does to a ray is grad n / n per unit arclength)
            scale = max(scale, float(np.max(np.abs(np.take(b, [3], axis=-2)))))
        diff = float(np.max(np.abs(aa - bb)))
        if diff > 0.0:
            worst = max(worst, diff / scale if scale > 0.0 else np.inf)
    return worst


def parity_relerr_elementwise(a, b):
    """The measure of rounds 1-3: max |a - b| / max(|b|, 1), element by element -- an ABSOLUTE 1e-9 for everything below 1 in
    magnitude (tighter than parity_relerr for a coordinate near 0, looser for p_x ~ 0.05).  Reported beside parity_relerr so
    that the 1e-9 claim reads the same across rounds; the tests assert both."""
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    if a.size == 0:
        return 0.0
    return float(np.max(np.abs(a - b) / np.maximum(np.abs(b), 1.0)))


def fan(scen, R_total, rank, world):
    lo, hi = SCEN[scen]["theta"]
    step = (hi - lo) / (R_total - 1)
    idx = np.arange(rank, R_total, world, dtype=np.float64)
    th = idx * step + lo
    if (R_total - 1) % world == rank:
        th[-1] = hi
    return th


def cpu_baseline(args, rb, budget_s, rays):
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
        if dt >= 0.4 * budget_s or used >= budget_s or R >= rays:
            break
        R = int(min(rays, max(2 * R, R * 0.6 * budget_s / max(dt, 1e-3))))
    return {"value": rate, "unit": "ray-steps/s", "cores": cores, "kind": "port",
            "sample": f"{args.scenario} op{args.method} fp64, {R} rays of the same fan, {r['steps']} ray-steps in "
                      f"{dt:.2f} s, OpenMP over rays, final-state mode"}


def parity_check(args, rb, batch, th, stride, step, max_size, lim):
    """The batch that was just timed against the oracle: every args.parity_stride-th ray -- step counts (exactly), d_ray and
    final state, and with a dense record every recorded row of those rays, read from the device copy.  fp64: 1e-9 relative
    (parity_relerr: per quantity, relative to that quantity's largest magnitude in the sample -- the measure of
    tests/test_gpu_parity.py); fp32 has no reference and is only reported."""
    from oracle import rt_oracle as O
    sc = SCEN[args.scenario]
    sub = slice(0, len(th), args.parity_stride)
    ths = np.ascontiguousarray(th[sub])
    fld = O.Field("vert_heterogeneous" if args.scenario == "anisotropy" else args.scenario, lim, rb.DELTA)
    want_rows = stride > 0
    rec_rows = batch.rec_rows
    o = O.trazar(fld, args.method, sc["gamma"], step, max_size, lim, sc["start"][0], sc["start"][1], ths,
                 record_stride=stride, rec_rows=rec_rows if want_rows else None,
                 nthreads=min(O.max_threads(), os.cpu_count() or 1))
    d = batch.d_ray()[:, sub]
    fin = batch.final()[:, sub]

    rel, rel_el = parity_relerr, parity_relerr_elementwise           # per-quantity scales; and rounds 1-3's element-wise measure
    steps_equal = bool(np.array_equal(d[2], o["d_ray"][2]))
    same = d[2] == o["d_ray"][2]
    err = max(rel(fin[:, same], o["final"][:, same]), rel(d[:2, same], o["d_ray"][:2, same]))
    err_el = max(rel_el(fin[:, same], o["final"][:, same]), rel_el(d[:2, same], o["d_ray"][:2, same]))
    out = {"rays": int(len(ths)), "every": args.parity_stride, "steps_equal": steps_equal,
           "rays_with_equal_steps": int(same.sum()), "max_rel_err": err, "max_rel_err_elementwise": err_el,
           "measures": "max_rel_err: per quantity, relative to that quantity's largest magnitude in the sample; "
                       "max_rel_err_elementwise: |a-b|/max(|b|,1); both must be below the tolerance",
           "oracle": "oracle/rt_oracle.c (kind: port)"}
    if want_rows:
        import torch
        s_dev = batch.device_tensors()["s_ray"][:, :, sub]                   # strided view of the device record
        s = s_dev.to(torch.float64).cpu().numpy()
        out["rows_compared"] = int(s.shape[0])
        out["rows_max_rel_err"] = rel(s[:, :, same], o["s_ray"][:, :, same])
        out["rows_max_rel_err_elementwise"] = rel_el(s[:, :, same], o["s_ray"][:, :, same])
        err_el = max(err_el, out["rows_max_rel_err_elementwise"])
        # rows past a ray's last written row must read 0 like the reference's np.zeros (RT_bench.py:802)
        last = (d[2] // stride).astype(np.int64)
        tail_ok = all(not s[int(last[k]) + 1:, :, k].any() for k in range(0, len(ths), max(1, len(ths) // 64)))
        out["rows_beyond_last_are_zero"] = bool(tail_ok)
        err = max(err, out["rows_max_rel_err"])
    tol = 1e-9 if args.dtype == "f64" else None
    out["tolerance"] = tol
    out["ok"] = bool(tol is None or (steps_equal and err < tol and err_el < tol and out.get("rows_beyond_last_are_zero", True)))
    return out


def trace_step(args, rb):
    c = rb.constants(SCEN[args.scenario]["choice"])
    if args.scenario == "fisheye":
        return 2 * np.pi / 303, rb.N * 304          # calibrated op6 step (RT_bench.py:1443, :1450)
    return rb.DELTA_S, int(np.ceil(c[4] / rb.DELTA_S) + 1)


def launcher_command(argv, gpus, port, python=None):
    """The N ranks of `python bench.py --gpus N ...` as one child command: torch.distributed.run, one process per GPU on this
    node, rendezvous on 127.0.0.1, the SAME bench.py arguments (argv = sys.argv[1:], untouched)."""
    return [python or sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={gpus}",
            "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + list(argv)


def free_port():
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def run_ranks(argv, gpus):
    """`python bench.py --gpus N` started plainly (no WORLD_SIZE in the environment), N > 1: start the N ranks as a CHILD
    process tree -- never an exec of this process, and before anything here has touched a GPU -- wait for it, print rank 0's
    ONE JSON line on stdout (everything else the ranks wrote goes to stderr) and return the child's exit status."""
    import subprocess
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC: what RCCL needs on this pool's host driver
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or gpus) // gpus)))
    cmd = launcher_command(argv, gpus, free_port())
    child = subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=None, env=env, text=True)
    line = None
    for out in child.stdout:
        if out.startswith('{"metric"'):
            line = out.rstrip("\n")
        else:
            sys.stderr.write(out)
    rc = child.wait()
    if line is not None:
        print(line, flush=True)
    elif rc == 0:
        sys.stderr.write("bench.py: the ranks exited 0 without a JSON line\n")
        rc = 1
    return rc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--rays", type=int, default=None,
                    help="weak scaling: this many rays per GPU (an N-times finer fan on N GPUs); makes the line's value the weak one")
    ap.add_argument("--total-rays", type=int, default=None,
                    help="strong scaling (the default, with 1 048 576): this many rays in total, split over the ranks")
    ap.add_argument("--no-weak", action="store_true",
                    help="N > 1, default (strong) run: skip the secondary weak record (1 048 576 rays per GPU on the same ranks)")
    ap.add_argument("--scenario", default="vert_heterogeneous", choices=sorted(SCEN))
    ap.add_argument("--method", type=int, default=None)
    ap.add_argument("--dtype", default="f64", choices=["f64", "f32"])
    ap.add_argument("--record", default="full", help="full (the reference's s_ray layout, default) | none | stride:N")
    ap.add_argument("--n-ray", action="store_true",
                    help="also keep n_ray rows (coef*n per row): internal to the reference's trazar (RT_bench.py:803), not among "
                         "its return values (:948), so off by default")
    ap.add_argument("--mode", default="auto", choices=["auto", "lane", "plain", "refill", "sliced"],
                    help="auto (default): rtmi_params' own default, RTMI_LAUNCH_AUTO -- the library picks between the "
                         "time-sliced and the plain schedule and keeps the faster on re-runs; plain (= lane): one block slot "
                         "per 256 rays to completion; refill: persistent waves with ballot/prefix lane refill; sliced: "
                         "persistent blocks advancing 256-ray bundles in time slices")
    ap.add_argument("--slice-steps", type=int, default=0, help="time-sliced schedule: DELTA_S steps per slice (0 = the library's 512)")
    ap.add_argument("--order", default="fan", choices=["fan", "shuffled"],
                    help="ray order inside the batch: the sorted fan, or a seeded random permutation of it")
    ap.add_argument("--refill-min", type=int, default=0)
    ap.add_argument("--sort", action="store_true", help="sort rays inside the batch by launch cell and angle (sort_rays)")
    ap.add_argument("--field-path", default="auto", choices=["auto", "lds", "global", "shared", "window"],
                    help="auto (default): the library's choice (rtmi_params.field_path 0); shared (= lds, its old name): the "
                         "wave-shared lookup -- a wave-uniform cell's polynomial through the scalar cache for the fast-form "
                         "methods, the LDS tile for the reference-order ones; window: as shared, the reference-order methods "
                         "through the scalar cache too (auto's choice from 131 072 rays on); global: every lane reads for itself")
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
    ap.add_argument("--gather-rows", type=int, default=64,
                    help="N>1 read-back: gather every this-many-th recorded row of every rank's trajectory to rank 0, device to "
                         "device (0 = only d_ray and the end points)")
    ap.add_argument("--emulate-world", type=int, default=0,
                    help="single GPU, no torch.distributed: run the shard rank --emulate-rank would own in a world of this size")
    ap.add_argument("--emulate-rank", type=int, default=0)
    ap.add_argument("--force-dist", action="store_true",
                    help="initialise torch.distributed even with one rank and run the N>1 collectives through it")
    ap.add_argument("--reference-order", action="store_true",
                    help="rtmi_params.reference_order = 1: op1/2/6/8 too in the reference's own operation order (op7 always is)")
    ap.add_argument("--fast-field", action="store_true",
                    help="rtmi_params.reference_order = 3: op7's reference-order step on the fused field lookup")
    ap.add_argument("--fused", action="store_true",
                    help="rtmi_params.reference_order = 2: fused forms wherever there is one -- op7 too (outside 1e-9 on interface)")
    ap.add_argument("--parity-stride", type=int, default=512, help="parity_check: every this-many-th ray (0 = skip)")
    args = ap.parse_args()
    if args.method is None:
        args.method = 11 if args.scenario == "anisotropy" else 6
    if args.mode == "lane":
        args.mode = "plain"

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # started as `python bench.py --gpus N`: this process becomes the launcher of its own N ranks (nothing here has
        # imported torch or touched a GPU yet); under torch.distributed.run WORLD_SIZE is set and this is one of the ranks
        sys.exit(run_ranks(sys.argv[1:], args.gpus))
    import torch
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        sys.exit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: the launcher's --nproc-per-node and --gpus must agree")
    if not torch.cuda.is_available():
        sys.exit("bench.py: no HIP device (raytracing_amd has no CPU path)")
    if args.all_on_device is not None:
        local = args.all_on_device
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    cdev = dev if args.backend == "nccl" else torch.device("cpu")     # where the collectives' tensors live
    use_dist = world > 1 or args.force_dist
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29531")
        kw = {} if world > 1 else dict(rank=0, world_size=1)
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=dev, **kw)
        else:
            dist.init_process_group("gloo", **kw)
    if args.emulate_world:
        if world != 1:
            sys.exit("bench.py: --emulate-world is a single-process rehearsal (run it with --gpus 1)")
        if not 0 <= args.emulate_rank < args.emulate_world:
            sys.exit("bench.py: --emulate-rank must be in [0, --emulate-world)")
    # the partition this process computes: its own rank of the real world, or one rank of an emulated one
    part_world, part_rank = (args.emulate_world, args.emulate_rank) if args.emulate_world else (world, rank)

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
    if args.rays is not None and args.total_rays:
        sys.exit("bench.py: --rays (weak: per GPU) and --total-rays (strong: in total) exclude each other")
    # The north-star run is the SAME 1 048 576-ray fan at every N (BASELINE.json's metric: "whole node ... 1M rays"),
    # so the default line is the strong split; --rays asks for the weak one.
    strong = args.rays is None
    R_total = (args.total_rays or 1048576) if strong else args.rays * part_world
    th = fan(args.scenario, R_total, part_rank, part_world)    # rank r owns rays r, r+world, ... of the R_total-ray fan
    R_local = len(th)
    if args.order == "shuffled":
        th = np.random.default_rng(1234 + rank).permutation(th)
    fld = rb.Field.build(args.scenario, lim, rb.DELTA, dtype)

    def make_batch(stride_, rec_rows_):
        return rb.Batch(fld, args.method, step, max_size, lim, sc["gamma"], th, sc["start"][0], sc["start"][1],
                        record_stride=stride_, rec_rows=rec_rows_, block_size=args.block,
                        launch_mode=args.mode, refill_min=args.refill_min, slice_steps=args.slice_steps,
                        field_path={"auto": 0, "global": 1, "lds": 2, "shared": 2, "window": 3}[args.field_path], sort_rays=args.sort,
                        lazy_clear=True,    # every pass re-runs the same launch conditions: same rows rewritten
                        keep_n_ray=args.n_ray, reference_order=3 if args.fast_field else 2 if args.fused else int(args.reference_order))

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
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    def one_pass(b):
        b.reset()
        if args.chunk <= 0:
            b.run()
            return
        # host-stepped: launches of `chunk` steps until no ray is live (checked every `group` launches: the check is a
        # reduction kernel + a host sync, which a real stepping caller would not pay per launch either)
        group = max(1, 256 // args.chunk)
        while True:
            b.step(args.chunk, group)          # `group` launches as one hipGraph (rtmi_step_repeat)
            if b.stats()["live_rays"] == 0:
                break

    # untimed passes: --warmup of them, at least AUTO_EXPLORE_RUNS with --mode auto -- RTMI_LAUNCH_AUTO times its two schedules
    # alternately on re-runs of a batch before it settles on one (rtmi_params.launch_mode), and that belongs to the warm-up,
    # not to the timed region
    untimed = max(args.warmup, rb.AUTO_EXPLORE_RUNS if args.mode == "auto" else 1)

    def timed_passes(b):
        """W untimed passes, then EXACTLY --steps passes between barrier + synchronize on both sides.  -> (seconds, stats
        before, stats after)."""
        for _ in range(untimed):
            one_pass(b)
        k0_ = b.stats()                      # kernel time and launches so far (rtmi_stats.*_total survive the resets)
        barrier()
        t0_ = time.perf_counter()
        for _ in range(args.steps):
            one_pass(b)
        barrier()
        dt_ = time.perf_counter() - t0_
        st_ = b.stats()                      # counters of the LAST pass (reset clears them), totals of all
        assert st_["live_rays"] == 0, "a pass ended with live rays"
        return dt_, k0_, st_

    dt, k0, st = timed_passes(batch)
    steps_per_pass = st["ray_steps"]
    # advance-kernel time per pass: the mean over the K timed passes themselves (HIP events on the batch's stream, folded
    # by the two stats calls around the timed region: no host sync per pass)
    kern_ms = (st["kernel_ms_total"] - k0["kernel_ms_total"]) / args.steps
    launches = max(int(round((st["launches_total"] - k0["launches_total"]) / args.steps)), 1)
    mode_used = st["launch_mode_used"] if args.chunk <= 0 else "plain"
    gathered = None
    parity_failed = False
    if use_dist:
        dt = rd.max_over_ranks(dt, cdev)
        total_steps = rd.sum_over_ranks(steps_per_pass, cdev) * args.steps
        # Read-back of the sharded layout (outside the timed region), device to device over the collective backend (nccl =
        # RCCL over xGMI): d_ray = (dist_real, dist_sim, last row) and the end points from zero-copy views of the library's
        # SoA state, and the trajectory itself -- every --gather-rows-th recorded row of s_ray[rows][6][R_local], a strided
        # view of the record in HBM (the whole 151 GB record of every rank does not fit one GPU; a rank's own record stays
        # where it is for the on-device consumers: metrics, isochrones, wavefronts).
        try:
            t_ = batch.device_tensors()
            loc = torch.stack((t_["dist_real"].double(), t_["dist_sim"].double(), t_["istep"].double(),
                               t_["x"].double(), t_["y"].double())).to(cdev)            # [5, R_local]
            Rmax = (R_total + world - 1) // world if not args.emulate_world else R_local
            R_all = R_local if args.emulate_world else R_total

            def gather_rays(t):                                                          # [.., R_local] -> [.., R_all] on rank 0
                if t.shape[-1] < Rmax:                                                   # ragged strong split: pad
                    t = torch.cat((t, torch.full(t.shape[:-1] + (Rmax - t.shape[-1],), float("nan"), dtype=t.dtype, device=t.device)), -1)
                g = [torch.empty_like(t) for _ in range(world)] if rank == 0 else None
                dist.gather(t.contiguous(), g, dst=0)
                return rd.interleave(g, R_all) if rank == 0 else None                    # ray k*world + r  <-  rank r, slot k
            allr = gather_rays(loc)
            traj = None
            if stride and args.gather_rows > 0 and "s_ray" in t_:
                sub = t_["s_ray"][::args.gather_rows].to(cdev)                           # [rows / g, 6, R_local]
                torch.cuda.synchronize()
                barrier()
                tg0 = time.perf_counter()
                traj = gather_rays(sub)
                torch.cuda.synchronize()
                tg = time.perf_counter() - tg0
            if rank == 0:
                assert int(allr[2].sum().item()) * args.steps == total_steps
                assert torch.isfinite(allr[3:5]).all()
                gathered = {"rays": int(allr.shape[1]), "backend": args.backend, "world": world,
                            "bytes_per_rank": int(loc.numel() * loc.element_size())}
                if traj is not None:
                    # rank 0's own rays sit at positions 0, world, 2*world, ...: the same bits as its local view
                    mine = traj[:, :, 0::world][:, :, :sub.shape[2]]
                    assert torch.equal(mine, sub), "gathered trajectory rows differ from rank 0's own record"
                    if args.dtype == "f64" and args.order == "fan" and not args.emulate_world:
                        # row 0 holds the launch angles (:826): in ray order they are the whole fan again
                        assert torch.equal(traj[0, 5], torch.from_numpy(fan(args.scenario, R_all, 0, 1)).to(traj.device)), \
                            "row 0 of the gathered trajectory is not the launch fan"
                    gathered["trajectory"] = {"rows": int(traj.shape[0]), "every": args.gather_rows, "shape": list(traj.shape),
                                              "bytes_per_rank": int(sub.numel() * sub.element_size()), "seconds": tg,
                                              "GB_per_s_into_rank0": sub.numel() * sub.element_size() * max(world - 1, 1) / tg / 1e9}
        except Exception as e:   # the timed result is already in hand; report and carry on
            print(f"bench.py: read-back gather failed: {e}", file=sys.stderr)
            gathered = {"error": str(e)}
    else:
        total_steps = steps_per_pass * args.steps

    if rank == 0:
        balg = alg_bytes_per_step(args.dtype, stride)
        ksec = kern_ms * 1e-3
        key = f"{args.scenario}:{R_local}:{args.record}{'+n_ray' if args.n_ray and stride else ''}:{args.dtype}:op{args.method}"
        if mode_used != "plain":
            key += ":" + mode_used            # profiles/traffic.json keeps the other launch modes under their own keys
        # the profiles were taken with every other option at its default: a run that changes one of them (ray order, sort,
        # field path, slice length, block size, refill threshold, a shard of a larger fan) has no profile of its own
        defaults = (args.order == "fan" and not args.sort and args.field_path == "auto" and args.slice_steps in (0, 512)
                    and args.block == 0 and args.refill_min == 0 and part_world == 1 and not args.reference_order and not args.fused and not args.fast_field)
        prof = {}
        try:
            prof = (json.load(open(os.path.join(ROOT, "profiles", "traffic.json"))).get(key, {}) or {}) if defaults else {}
        except Exception:
            prof = {}
        if not isinstance(prof, dict):        # round-1 format: a bare byte count
            prof = {"hbm_bytes": prof, "source": "profiles/r01_f_head_pmc_summary.txt"}
        # the profile describes a BUILD: its counters are borrowed only while the kernel that just ran is the one that was
        # profiled -- same kernel, same register count (profile_summary.py records both; entries of before round 4 carry neither
        # and are taken as they are)
        kernel_name = {"refill": "k_trace_refill", "sliced": "k_advance_sliced"}.get(mode_used, "k_advance")
        stale = None
        if prof and (prof.get("vgprs") not in (None, st["vgprs"]) or prof.get("kernel") not in (None, kernel_name)):
            stale = (f"{prof.get('source')} was measured on {prof.get('kernel')} with {prof.get('vgprs')} VGPRs; this build ran "
                     f"{kernel_name} with {st['vgprs']}: not used")
            prof = {}
        measured = prof.get("hbm_bytes") if args.chunk <= 0 else None
        model = min_hbm_bytes(args.dtype, stride, steps_per_pass, R_local, args.method, args.n_ray)
        hbm_bytes = measured if measured else model
        hbm = {"achieved": hbm_bytes / ksec / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
               "frac": hbm_bytes / ksec / 1e9 / HBM_PEAK_GBS,
               "bytes_per_pass": hbm_bytes,
               "bytes_source": prof.get("source") if measured else "model: recorded rows x 6 values (7 with n_ray) + ray state in and out"}
        if stride:   # secondary yardstick for the row stream: what a write-only kernel reaches on this memory system
            hbm["write_only_ceiling"] = {"peak": HBM_FILL_GBS, "unit": "GB/s", "frac": hbm_bytes / ksec / 1e9 / HBM_FILL_GBS,
                                         "source": "profiles/r02_f_hbm_write_ceiling.txt (tools/hbm_fill.hip)"}
        roof = dict(hbm, bound="hbm", traffic=measured, traffic_source=prof.get("source") if measured else None,
                    traffic_key=key if measured else None)
        if stale:
            roof["traffic_stale"] = stale
        roof.pop("write_only_ceiling", None)
        # vector-ALU issue time from the profiled instruction counts: a 64-lane fp64 instruction holds a SIMD for 4 cycles
        # (16 lanes/clk), any other VALU instruction for 2 (SIMD-32); against 1024 SIMDs at the 2.4 GHz peak clock
        if measured and prof.get("valu_insts"):
            f64 = prof.get("valu_f64_insts", 0.0)
            cyc = 4.0 * f64 + 2.0 * (prof["valu_insts"] - f64)
            frac = cyc / SIMDS / PEAK_CLOCK_HZ / ksec
            valu = {"achieved": cyc / ksec / 1e9, "peak": SIMDS * PEAK_CLOCK_HZ / 1e9, "unit": "G SIMD-cycles/s", "frac": frac,
                    "valu_wave_insts_per_pass": prof.get("valu_insts"), "fp64_wave_insts_per_pass": f64,
                    "salu_wave_insts_per_pass": prof.get("salu_insts")}
            roof["valu_issue"] = valu
            if frac > roof["frac"]:
                roof.update(bound="valu", achieved=valu["achieved"], peak=valu["peak"], unit=valu["unit"], frac=frac)
        roof.update(kernel=kernel_name, kernel_ms=kern_ms / launches,
                    kernel_ms_per_pass=kern_ms, launches_per_pass=launches, ray_steps_per_pass=int(steps_per_pass),
                    vgprs=st["vgprs"], hbm=hbm,
                    alg_model={"bytes_per_ray_step": balg, "GB_per_s": balg * steps_per_pass / ksec / 1e9,
                               "note": "SURVEY.md 8d accounting (state in/out + 36-coefficient gather per ray-step); not a "
                                       "bound for this kernel: state stays in registers for all steps of a ray and the "
                                       "lookup reads a wave-uniform cell's coefficients through the scalar cache, so these bytes never reach HBM"})
        out = {
            "metric": "ray-steps/sec (whole node) on vert_heterogeneous, 1M rays; % HBM roofline",
            "value": total_steps / dt, "unit": "ray-steps/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "untimed_passes": untimed, "ms_per_step": 1e3 * dt / args.steps, "higher_is_better": True,
            "scaling": "strong" if strong else "weak", "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": f"{args.scenario}, {R_total} rays in total ({R_local} on rank 0; fan linspace over {R_total} "
                                   f"rays, interleaved across ranks), op{args.method}, DELTA_S={step:.12g}, "
                                   f"box={tuple(float(v) for v in lim)}, record={args.record}",
                       "rays_total": R_total, "rays_rank0": R_local, "ray_steps_per_pass_rank0": int(steps_per_pass),
                       "method": f"op{args.method}", "record": args.record, "n_ray_rows": bool(args.n_ray and stride), "launch_mode": args.mode, "launch_mode_used": mode_used, "auto_fallbacks": int(st["auto_fallbacks"]), "retraced": int(st["retraced"]), "retrace_overflow": int(st["retrace_overflow"]), "dispatch_first": int(st["dispatch_first"]), "reference_order": 3 if args.fast_field else 2 if args.fused else int(args.reference_order),
                       "ray_order": args.order, "sort_rays": bool(args.sort), "field_path": args.field_path,
                       "steps_per_launch": args.chunk or "all", "parallelism": f"ray-shard x{world}"},
            "roofline": roof,
        }
        out["config"]["auto_exploration"] = st["auto_exploration"]     # RTMI_LAUNCH_AUTO: both schedules' kernel ms, and the one kept
        if args.emulate_world:
            out["config"]["emulated"] = {"world": part_world, "rank": part_rank,
                                         "note": "one rank's shard of the N-way split, run alone on one GPU"}
        if use_dist:
            out["config"]["dist"] = {"backend": args.backend, "world_size": world, "forced": bool(args.force_dist and world == 1),
                                     "gather": gathered}
        if world == 1 and args.cpu_seconds > 0:
            out["cpu_baseline"] = cpu_baseline(args, rb, args.cpu_seconds, R_local)
        if args.parity_stride > 0 and args.chunk <= 0:
            out["parity_check"] = parity_check(args, rb, batch, th, stride, step, max_size, lim)
            parity_failed = not out["parity_check"]["ok"]
    batch.close()

    # The secondary record of the default N > 1 line: the WEAK configuration on the same ranks -- 1 048 576 rays per GPU of an
    # N-times finer fan, everything else as above -- timed by the same barrier-bracketed loop.  (Weak scaling has nothing to
    # lose here: no collective in the timed region, every GPU does the single-GPU job; it is reported so that the driver's
    # per-GPU figure can be read beside the strong split's.)
    if strong and world > 1 and not args.no_weak and args.rays is None and not args.emulate_world:
        weak = {"scaling": "weak", "rays_per_gpu": 1048576, "rays_total": 1048576 * world}
        try:
            th_w = fan(args.scenario, 1048576 * world, rank, world)
            if args.order == "shuffled":
                th_w = np.random.default_rng(1234 + rank).permutation(th_w)
            th = th_w                                   # make_batch reads `th`
            bw = make_batch(stride, rec_rows)
            dt_w, k0_w, st_w = timed_passes(bw)
            dt_w = rd.max_over_ranks(dt_w, cdev)
            steps_w = rd.sum_over_ranks(st_w["ray_steps"], cdev) * args.steps
            weak.update(value=steps_w / dt_w, unit="ray-steps/s", ms_per_step=1e3 * dt_w / args.steps, steps=args.steps,
                        kernel_ms_per_pass_rank0=(st_w["kernel_ms_total"] - k0_w["kernel_ms_total"]) / args.steps,
                        launch_mode_used=st_w["launch_mode_used"], auto_exploration=st_w["auto_exploration"])
            bw.close()
        except Exception as e:       # the primary (strong) result stands; say what happened to the secondary one
            weak["error"] = str(e)
            if use_dist:             # keep the ranks' collectives paired: the failing rank still joins both reductions
                try:
                    rd.max_over_ranks(0.0, cdev); rd.sum_over_ranks(0, cdev)
                except Exception:
                    pass
        if rank == 0:
            out["weak"] = weak
    if rank == 0:
        print(json.dumps(out), flush=True)
    fld.close()
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()
    if parity_failed:
        sys.exit("bench.py: parity_check failed -- the timed batch differs from the oracle (see the JSON line)")


if __name__ == "__main__":
    main()
