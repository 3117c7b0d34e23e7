"""GPU: rtmi_shard (include/rtmi.h) -- one call's rays over several devices from one process, read back device to device.
One MI355X here: RCCL with a single rank (ncclCommInitAll + ncclGather really run), and the N-way split rehearsed by listing
device 0 several times (peer-copy transport).  Either way the answers are the bits of one unsharded batch."""
import numpy as np
import pytest

from conftest import LIMITS

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def rb():
    from raytracing_amd import rt_bench
    return rt_bench


def _reference(rb, scen, m, step, ms, gam, th, x0, y0, stride, rows, dtype=0):
    F = rb.Field.build(scen, LIMITS[scen], rb.DELTA, dtype)
    b = rb.Batch(F, m, step, ms, LIMITS[scen], gam, th, x0, y0, record_stride=stride, rec_rows=rows, keep_n_ray=False)
    b.run()
    out = (b.d_ray(), b.final(), b.rows() if stride else None, b.stats()["ray_steps"])
    b.close(); F.close()
    return out


@pytest.mark.parametrize("devices,transport,expect", [([0], "auto", "rccl"), ([0], "copy", "copy"), ([0, 0], "auto", "copy"),
                                                      ([0, 0, 0], "copy", "copy")])
def test_shard_equals_one_batch(devices, transport, expect, rb):
    scen, m, gam = "vert_heterogeneous", 6, 1
    th = np.linspace(0, np.pi / 2, 1001)                 # 1001 rays: ragged three-way split (334 + 334 + 333)
    ms, rows = 3100, 3100
    d0, f0, s0, steps0 = _reference(rb, scen, m, rb.DELTA_S, ms, gam, th, -2.0, -2.0, 1, rows)
    sh = rb.Shard(scen, m, rb.DELTA_S, ms, LIMITS[scen], gam, th, -2.0, -2.0, devices, record_stride=1, rec_rows=rows, transport=transport)
    sh.run()
    info = sh.info()
    assert info["transport"] == expect and info["ndev"] == len(devices) and info["R"] == 1001
    assert info["ray_steps"] == steps0 and info["live_rays"] == 0 and info["auto_fallbacks"] == 0
    assert np.array_equal(sh.d_ray(), d0) and np.array_equal(sh.final(), f0)
    assert np.array_equal(sh.rows(0, rows), s0)
    assert np.array_equal(sh.rows(5, 40, every=64), s0[5::64][:40])
    sh.reset(); sh.run()                                  # the benchmark loop's re-run (RT_bench.py:1520-1523)
    assert np.array_equal(sh.d_ray(), d0)
    sh.close()


def _gpus():
    from raytracing_amd import _lib
    import ctypes as C
    n = C.c_int(0)
    _lib.check(_lib.lib().rtmi_device_count(C.byref(n)))
    return n.value


@pytest.mark.parametrize("devices,transport,expect", [([0, 1], "rccl", "rccl"), ([0, 1], "copy", "copy"), ([1, 0], "auto", "rccl"),
                                                      ([0, 1, 2], "auto", "rccl"), ([0, 1, 2, 3], "copy", "copy")])
def test_shard_on_distinct_devices(devices, transport, expect, rb):
    """The shard split over DISTINCT GPUs -- ncclCommInitAll with several communicators, ncclGather between them,
    hipMemcpyPeerAsync between devices, k_interleave on gathered remote blocks -- against one batch on device 0, bit for bit;
    and the caller's current device is the one it was before every call (include/rtmi.h).  Runs where the host has the GPUs (the
    driver's multi-GPU node); a one-GPU box skips it."""
    if _gpus() <= max(devices):
        pytest.skip(f"needs {max(devices) + 1} GPUs, this host has {_gpus()}")
    import torch
    from raytracing_amd import _lib
    _lib.check(_lib.lib().rtmi_set_device(0))
    scen, m, gam = "vert_heterogeneous", 6, 1
    th = np.linspace(0, np.pi / 2, 1001)                 # ragged splits
    ms, rows = 3100, 3100
    d0, f0, s0, steps0 = _reference(rb, scen, m, rb.DELTA_S, ms, gam, th, -2.0, -2.0, 1, rows)
    cur = torch.cuda.current_device()
    sh = rb.Shard(scen, m, rb.DELTA_S, ms, LIMITS[scen], gam, th, -2.0, -2.0, devices, record_stride=1, rec_rows=rows, transport=transport)
    assert torch.cuda.current_device() == cur
    sh.run()
    info = sh.info()
    assert torch.cuda.current_device() == cur
    assert info["transport"] == expect and info["ndev"] == len(devices) and info["ray_steps"] == steps0 and info["live_rays"] == 0
    assert np.array_equal(sh.d_ray(), d0) and np.array_equal(sh.final(), f0)
    assert np.array_equal(sh.rows(0, rows), s0) and np.array_equal(sh.rows(5, 40, every=64), s0[5::64][:40])
    assert torch.cuda.current_device() == cur
    sh.reset(); sh.run()
    assert np.array_equal(sh.d_ray(), d0)
    # a second shard on the same devices (peer access already enabled) and an ordinary batch afterwards on the caller's device
    sh2 = rb.Shard(scen, m, rb.DELTA_S, ms, LIMITS[scen], gam, th[:333], -2.0, -2.0, devices, record_stride=0, transport="copy")
    sh2.run()
    assert np.array_equal(sh2.d_ray(), d0[:, :333])
    sh2.close(); sh.close()
    assert torch.cuda.current_device() == cur
    d1, *_ = _reference(rb, scen, m, rb.DELTA_S, ms, gam, th[:64], -2.0, -2.0, 0, 0)
    assert np.array_equal(d1, d0[:, :64])


def test_shard_leaves_the_callers_device_alone(rb):
    """One GPU: every rtmi_shard_* entry restores the calling thread's current device (a handle of the rest of the ABI made before
    still works after), and a second shard right after the first does not trip over a stale HIP error."""
    th = np.linspace(0, np.pi / 2, 257)
    F = rb.Field.build("vert_heterogeneous")
    b = rb.Batch(F, 6, rb.DELTA_S, 3100, LIMITS["vert_heterogeneous"], 1, th, -2.0, -2.0, record_stride=0)
    for _ in range(2):
        sh = rb.Shard("vert_heterogeneous", 6, rb.DELTA_S, 3100, LIMITS["vert_heterogeneous"], 1, th, -2.0, -2.0, [0, 0], record_stride=0)
        sh.run()
        d = sh.d_ray()
        sh.info(); sh.close()
    b.run()                                   # a batch made before the shard calls: its field's device is still current
    assert np.array_equal(b.d_ray(), d)
    b.close(); F.close()


def test_shard_other_methods_and_precisions(rb):
    rng = np.random.default_rng(3)
    for scen, m, gam, dtype, th in (("anisotropy", 11, 3, 0, np.linspace(0, np.pi / 2, 130)),
                                    ("interface", 7, 1, 0, np.linspace(0.2, 1.5, 97)),
                                    ("vert_heterogeneous", 6, 1, 1, rng.permutation(np.linspace(0, np.pi / 2, 777)))):
        key = "vert_heterogeneous" if scen == "anisotropy" else scen
        d0, f0, s0, _ = _reference(rb, key, m, rb.DELTA_S, 1500, gam, th, -2.0, -2.0, 8, 0, dtype)
        sh = rb.Shard(key, m, rb.DELTA_S, 1500, LIMITS[key], gam, th, -2.0, -2.0, [0, 0], record_stride=8, dtype=dtype)
        sh.run()
        assert np.array_equal(sh.d_ray(), d0) and np.array_equal(sh.final(), f0), scen
        assert np.array_equal(sh.rows(0, s0.shape[0]), s0), scen
        sh.close()


def test_shard_argument_errors(rb):
    from raytracing_amd._lib import RtmiError
    th = np.linspace(0, 1, 8)
    with pytest.raises(RtmiError, match="device index"):
        rb.Shard("vert_heterogeneous", 6, rb.DELTA_S, 100, LIMITS["vert_heterogeneous"], 1, th, -2.0, -2.0, [0, 99])
    with pytest.raises(RtmiError, match="distinct"):
        rb.Shard("vert_heterogeneous", 6, rb.DELTA_S, 100, LIMITS["vert_heterogeneous"], 1, th, -2.0, -2.0, [0, 0], transport="rccl")
    with pytest.raises(RtmiError, match="fewer rays"):
        rb.Shard("vert_heterogeneous", 6, rb.DELTA_S, 100, LIMITS["vert_heterogeneous"], 1, th[:1], -2.0, -2.0, [0, 0])
