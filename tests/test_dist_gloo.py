"""CPU, world_size 2, gloo: the N>1 plumbing of bench.py / raytracing_amd.dist -- ray partition, the read-back
gather and the timing reductions.  The per-rank propagation is stood in for by the oracle (this is the checker
exercising the plumbing; on the GPU box the same code paths carry librtmi results over RCCL)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import LIMITS, ROOT
from raytracing_amd import dist as rd

R_TOTAL = 37      # odd on purpose: ragged blocks


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


def _trace(th):
    from oracle import rt_oracle as O
    F = O.Field("vert_heterogeneous", LIMITS["vert_heterogeneous"], 0.017644349415748446)
    r = O.trazar(F, 6, 1, 0.002646652412362267, 600, LIMITS["vert_heterogeneous"], -2.0, -2.0, th, record_stride=0)
    return np.concatenate([r["final"], r["d_ray"]], axis=0), r["steps"]     # [12, R]


def _worker(rank, world, port, tmp):
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        # --- block partition + ragged gather
        th = rd.fan_shard(0.0, np.pi / 2, R_TOTAL, rank, world)
        full = np.linspace(0.0, np.pi / 2, R_TOTAL)
        lo, hi = rd.shard_range(R_TOTAL, rank, world)
        assert np.array_equal(th, full[lo:hi])
        res, steps = _trace(th)
        g = rd.gather_blocks(torch.from_numpy(res), R_TOTAL, dst=0)
        total = rd.sum_over_ranks(steps)
        tmax = rd.max_over_ranks(1.0 + rank)
        assert tmax == float(world)
        # --- interleaved partition used by bench.py
        import bench
        thi = bench.fan("vert_heterogeneous", R_TOTAL + 1, rank, world)       # 38 rays: 19 per rank
        assert np.array_equal(thi, np.linspace(0.0, np.pi / 2, R_TOTAL + 1)[rank::world])
        resi, _ = _trace(thi)
        ti = torch.from_numpy(resi)
        gl = [torch.empty_like(ti) for _ in range(world)] if rank == 0 else None
        dist.gather(ti, gl, dst=0)
        # --- strong-scaling split of bench.py --total-rays: the SAME 37-ray fan, ragged (19 + 18), padded gather
        ths = bench.fan("vert_heterogeneous", R_TOTAL, rank, world)
        assert np.array_equal(ths, full[rank::world])
        ress, _ = _trace(ths)
        ts = torch.from_numpy(ress)
        Rmax = (R_TOTAL + world - 1) // world
        if ts.shape[1] < Rmax:
            ts = torch.cat((ts, torch.full((ts.shape[0], Rmax - ts.shape[1]), float("nan"), dtype=ts.dtype)), 1)
        gs = [torch.empty_like(ts) for _ in range(world)] if rank == 0 else None
        dist.gather(ts.contiguous(), gs, dst=0)
        # --- the agreement before a collective (trazar_sharded: a failed rank must not leave the others in the gather) and the
        #     slot-order -> caller's-order map of a sort_rays batch
        assert rd.agree_ok(True) is True
        assert rd.agree_ok(rank != 1) is False                              # rank 1 "failed": every rank learns it
        perm = torch.tensor([2, 0, 3, 1], dtype=torch.int32)
        slots = torch.arange(8.0).reshape(2, 4)                             # slot k holds the caller's ray perm[k]
        assert torch.equal(rd._to_callers_order(slots, perm), torch.tensor([[1.0, 3.0, 0.0, 2.0], [5.0, 7.0, 4.0, 6.0]]))
        assert rd._to_callers_order(slots, None) is slots
        if rank == 0:
            whole, wsteps = _trace(full)
            assert np.array_equal(rd.interleave(gs, R_TOTAL).numpy(), whole)        # strong split: same bits, ray order
            assert g.shape == (12, R_TOTAL) and np.array_equal(g.numpy(), whole)    # bit-identical to unsharded
            assert total == wsteps
            wi, _ = _trace(np.linspace(0.0, np.pi / 2, R_TOTAL + 1))
            assert np.array_equal(rd.interleave(gl, R_TOTAL + 1).numpy(), wi)
            open(os.path.join(tmp, "ok"), "w").write("ok")
        else:
            assert g is None
        dist.barrier()
    finally:
        dist.destroy_process_group()


def test_shard_range_covers_everything():
    for R in (1, 7, 64, 1000, 1048576):
        for w in (1, 2, 3, 8):
            edges = [rd.shard_range(R, r, w) for r in range(w)]
            assert edges[0][0] == 0 and edges[-1][1] == R
            assert all(edges[i][1] == edges[i + 1][0] for i in range(w - 1))
            sizes = [b - a for a, b in edges]
            assert max(sizes) - min(sizes) <= 1


@pytest.mark.timeout(300)
def test_two_rank_shard_gather_gloo(tmp_path):
    mp.spawn(_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    assert (tmp_path / "ok").exists()
