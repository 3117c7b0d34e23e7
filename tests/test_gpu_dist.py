"""GPU, two processes on one MI355X (gloo for the rendezvous and the gather; RCCL needs one GPU per rank and runs on the
driver's multi-GPU node): raytracing_amd.dist.trazar_sharded -- one trazar() call's rays split over the ranks -- gives the
unsharded call's arrays bit for bit."""
import os
import socket
import sys

import numpy as np
import pytest
import torch.multiprocessing as mp

from conftest import ROOT

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


def _worker(rank, world, port, tmp):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    torch.cuda.init()                                   # torch's HIP runtime first (raytracing_amd/_lib.py)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from raytracing_amd import dist as rd
        from raytracing_amd import rt_bench as rb
        cases = [("vert_heterogeneous", "3", rb.op6, np.linspace(0, np.pi / 2, 1001), 16, 1),      # ragged: 501 + 500
                 ("anisotropy", "4", rb.op11, np.linspace(0, np.pi / 2, 64), None, 3),
                 ("interface", "1", rb.op6, None, "full", 1)]                                         # the 42-ray preset
        for scen, choice, op, th, record, gam in cases:
            rb.gamma = gam
            res = rd.trazar_sharded(op, scen, False, rb.DELTA_S, 91, choice, thetas=th, record=record, device=0)
            if rank == 0:
                fld = rb.Field.build(scen)
                z, grd = rb.FieldSpline(fld, "n"), (rb.FieldSpline(fld, "dy"), rb.FieldSpline(fld, "dx"))
                ref = rb.trazar(op, z, grd, False, rb.DELTA_S, 91, choice, thetas=th, record=record)
                fld.close()
                assert np.array_equal(res[1], ref[1]), scen                       # d_ray
                assert np.array_equal(res[3], ref[3]), scen                       # errors (interface: the 42 exit-angle errors)
                if record is not None:
                    assert res[0].shape == ref[0].shape and np.array_equal(res[0], ref[0]), scen
                else:
                    assert res[0] is None
            else:
                assert res is None
        # a rank without rays (more ranks than rays) contributes padding, the gather still answers in ray order
        one = np.array([0.7])
        res = rd.trazar_sharded(rb.op6, "vert_heterogeneous", False, rb.DELTA_S, 91, "3", thetas=one, record=64, device=0)
        if rank == 0:
            fld = rb.Field.build("vert_heterogeneous")
            ref = rb.trazar(rb.op6, rb.FieldSpline(fld, "n"), (rb.FieldSpline(fld, "dy"), rb.FieldSpline(fld, "dx")), False, rb.DELTA_S, 91, "3",
                            thetas=one, record=64)
            fld.close()
            assert res[0].shape == ref[0].shape and np.array_equal(res[0], ref[0]) and np.array_equal(res[1], ref[1])
        # a rank whose local trace fails: every rank raises instead of one of them waiting in the gather forever
        real = rb.trazar
        if rank == 1:
            def broken(*a, **k):
                raise rb._lib.RtmiError(-3, "injected failure")
            rb.trazar = broken
        try:
            rd.trazar_sharded(rb.op6, "vert_heterogeneous", False, rb.DELTA_S, 91, "3", thetas=np.linspace(0, 1, 8), device=0)
            raised = False
        except RuntimeError as e:
            raised = "local trace failed" in str(e)
        finally:
            rb.trazar = real
        assert raised, f"rank {rank} did not raise"
        if rank == 0:
            open(os.path.join(tmp, "ok"), "w").write("ok")
        dist.barrier()
    finally:
        dist.destroy_process_group()


def _worker_rccl(rank, world, port, tmp):
    """Backend nccl (RCCL), one rank per GPU: trazar_sharded's device-to-device payload -- zero-copy views of the batch's HBM arrays
    through dist.gather -- against the plain call.  world 1 on a one-GPU box (one GPU carries one RCCL rank); world >= 2 -- real
    peers over xGMI, ragged splits, a rank-0 gather of other devices' blocks -- where the host has the GPUs."""
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    torch.cuda.init()
    torch.cuda.set_device(rank)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", rank))
    try:
        from raytracing_amd import dist as rd
        from raytracing_amd import rt_bench as rb
        rng = np.random.default_rng(5)
        cases = [("vert_heterogeneous", "3", rb.op6, np.linspace(0, np.pi / 2, 1001), "full", rb.F64),
                 ("vert_heterogeneous", "3", rb.op6, rng.permutation(np.linspace(0, np.pi / 2, 4096)), 16, rb.F64),   # sort_rays kicks in: perm
                 ("vert_heterogeneous", "3", rb.op7, np.linspace(0, np.pi / 2, 300), None, rb.F64),
                 ("vert_heterogeneous", "3", rb.op6, np.linspace(0, np.pi / 2, 777), 8, rb.F32),
                 ("interface", "1", rb.op6, None, "full", rb.F64)]
        for scen, choice, op, th, record, dtype in cases:
            res = rd.trazar_sharded(op, scen, False, rb.DELTA_S, 91, choice, thetas=th, record=record, dtype=dtype, device=rank)
            if rank != 0:
                assert res is None
                continue
            fld = rb.Field.build(scen, dtype=dtype)
            z, grd = rb.FieldSpline(fld, "n"), (rb.FieldSpline(fld, "dy"), rb.FieldSpline(fld, "dx"))
            ref = rb.trazar(op, z, grd, False, rb.DELTA_S, 91, choice, thetas=th, record=record)
            fld.close()
            assert np.array_equal(res[1], ref[1]) and np.array_equal(res[3], ref[3]), scen
            if record is not None:
                assert res[0].shape == ref[0].shape and np.array_equal(res[0], ref[0]), (scen, record)
            else:
                assert res[0] is None
        if rank == 0:
            open(os.path.join(tmp, "ok"), "w").write("ok")
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(600)
def test_trazar_sharded_device_gather_over_rccl(tmp_path):
    mp.spawn(_worker_rccl, args=(1, _free_port(), str(tmp_path)), nprocs=1, join=True)
    assert (tmp_path / "ok").exists()


def _gpus():
    import torch
    return torch.cuda.device_count()


@pytest.mark.timeout(600)
@pytest.mark.parametrize("world", [2, 4])
def test_trazar_sharded_over_rccl_on_distinct_gpus(world, tmp_path):
    """The N > 1 RCCL path proper: one rank per GPU, ncclGather between real peers.  Runs where the host has the GPUs (the
    driver's multi-GPU node); a one-GPU box skips it."""
    if _gpus() < world:
        pytest.skip(f"needs {world} GPUs, this host has {_gpus()}")
    mp.spawn(_worker_rccl, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    assert (tmp_path / "ok").exists()


@pytest.mark.timeout(600)
def test_trazar_sharded_two_ranks_one_gpu(tmp_path):
    mp.spawn(_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    assert (tmp_path / "ok").exists()
