"""GPU: the HIP path (through the C ABI, via raytracing_amd.rt_bench) against the oracle and against the
golden vectors captured from the reference.  Tolerances (fp64):
  * field build / n_gradient: 1e-13 of the coefficient scale (pure arithmetic, FMA contraction only);
  * trajectories, non-golden methods: 1e-9 relative (north_star); measured ~1e-13;
  * golden-section methods (op5/9/10/11): a flipped cost comparison would move that step's angle by <= 6e-8
    (SURVEY.md section 7); they run in the reference's own operation order (rt_exact.h) and are held to 1e-9 on
    every ray here and to the oracle's exact bits in tests/test_gpu_exact.py.
"""
import os

import numpy as np
import pytest

from bench import parity_relerr, parity_relerr_elementwise
from conftest import LIMITS, golden, sub_rows, traj_fixtures, traj_inputs

pytestmark = pytest.mark.gpu

REL = 1e-9


@pytest.fixture(scope="module")
def rb():
    from raytracing_amd import rt_bench
    n = __import__("ctypes").c_int()
    from raytracing_amd import _lib
    _lib.check(_lib.lib().rtmi_device_count(n))
    assert n.value >= 1, "no HIP device"
    return rt_bench


@pytest.fixture(scope="module")
def gpu_fields(rb):
    cache = {}

    def get(scen, dtype=0):
        key = ("vert_heterogeneous" if scen == "anisotropy" else scen, dtype)
        if key not in cache:
            cache[key] = rb.Field.build(key[0], LIMITS[key[0]], rb.DELTA, dtype)
        return cache[key]
    yield get
    for f in cache.values():
        f.close()


def relerr(a, b, floor=None):
    """Arrays [.., quantity, ray] (final state, d_ray, recorded rows): bench.parity_relerr -- relative to each quantity's own
    largest magnitude over the fixture, so that 1e-9 means 1e-9 of p_x ~ 0.05 or T ~ 0.4 too -- AND the element-wise
    |a - b| / max(|b|, 1) of rounds 1-3 (an absolute 1e-9 below 1): the larger of the two, so a bound holds under both.  With
    `floor` (flat vectors of mixed quantities): element-wise |a - b| / max(|b|, floor)."""
    if floor is not None:
        return np.max(np.abs(a - b) / np.maximum(np.abs(b), floor))
    # BOTH measures (the larger): per quantity relative to that quantity's scale, and rounds 1-3's |a - b| / max(|b|, 1)
    return max(parity_relerr(a, b), parity_relerr_elementwise(a, b))


# ------------------------------------------------------------------ field (a1-a5)
@pytest.mark.parametrize("scen", ["interface", "fisheye", "vert_heterogeneous"])
def test_field_build_parity(scen, rb, gpu_fields, oracle_fields):
    F, OF = gpu_fields(scen), oracle_fields(scen)
    assert (F.qx, F.qy) == (OF.qx, OF.qy)
    x, y, Z, cdy, cdx = F.arrays()
    ox, oy, oZ, ocdy, ocdx = OF.arrays()
    assert np.array_equal(x, ox) and np.array_equal(y, oy)
    # samples (interface: numpy's SVML exp restated on both sides), np.gradient stencil, FITPACK's Givens QR: the same bits
    assert np.array_equal(Z, oZ) and np.array_equal(cdy, ocdy) and np.array_equal(cdx, ocdx)
    g = golden(f"field_{scen}")                                        # and straight against the reference
    qy, qx = Z.shape
    for name, arr in (("Z", Z), ("cdy", cdy), ("cdx", cdx)):
        assert np.array_equal(arr[qy // 2:qy // 2 + 8, qx // 2:qx // 2 + 8], g[name + "_mid"])


@pytest.mark.parametrize("scen", ["interface", "fisheye", "vert_heterogeneous"])
def test_n_gradient_parity(scen, rb, gpu_fields, oracle_fields):
    g = golden(f"field_{scen}")
    F = gpu_fields(scen)
    n, gx, gy = F.n_gradient(g["px"], g["py"])
    _, _, Z, cdy, cdx = oracle_fields(scen).arrays()
    assert np.abs(n - g["n"]).max() <= 1e-14 * np.abs(Z).max()
    gscale = max(np.abs(cdx).max(), np.abs(cdy).max())
    assert np.abs(gx - g["gx"]).max() <= 1e-13 * gscale
    assert np.abs(gy - g["gy"]).max() <= 1e-13 * gscale
    # reference-style scalar call surface
    z, grd = rb.FieldSpline(F, "n"), (rb.FieldSpline(F, "dy"), rb.FieldSpline(F, "dx"))
    nn, gg = rb.n_gradient(np.array((g["px"][20], g["py"][20])), grd, z)
    assert nn == n[20] and gg[0] == gx[20] and gg[1] == gy[20]
    assert z(g["py"][20], g["px"][20]).shape == (1, 1)


def test_field_clamps_outside_grid(rb, gpu_fields, oracle_fields):
    F, OF = gpu_fields("vert_heterogeneous"), oracle_fields("vert_heterogeneous")
    px = np.array([100.0, -100.0, 8.0, -5.0, 7.9999, 0.0]); py = np.array([100.0, -100.0, 4.0, -5.5, 3.9999, 50.0])
    a = F.n_gradient(px, py); b = OF.n_gradient(px, py)
    for u, v in zip(a, b):
        assert np.abs(u - v).max() <= 1e-13 * max(np.abs(b[0]).max(), np.abs(b[2]).max())


def test_from_samples_equals_build(rb, gpu_fields):
    F = gpu_fields("fisheye")
    x, y, Z, cdy, cdx = F.arrays()
    G = rb.Field.from_samples(x, y, Z)
    _, _, Z2, cdy2, cdx2 = G.arrays()
    assert np.array_equal(Z, Z2) and np.array_equal(cdy, cdy2) and np.array_equal(cdx, cdx2)
    G.close()
    from raytracing_amd._lib import RtmiError
    xb = x.copy(); xb[5] += 1e-9
    with pytest.raises(RtmiError, match="linspace"):
        rb.Field.from_samples(xb, y, Z)


# ------------------------------------------------------------------ one step (a6-a14), every method
@pytest.mark.parametrize("m", range(1, 12))
def test_single_step_parity(m, rb, gpu_fields):
    g = golden("step_methods")
    F = gpu_fields("vert_heterogeneous")
    st, hist, ref = g[f"st{m}"], g[f"hist{m}"], g[f"out{m}"]
    R = st.shape[0]
    gam = 3 if m >= 10 else 1
    b = rb.Batch(F, m, float(g["step"]), 1 << 20, (-1e300, 1e300, -1e300, 1e300), gam, st[:, 2], st[:, 0], st[:, 1],
                 record_stride=0)
    state9 = np.zeros((9, R)); state9[:6] = st[:, :6].T
    b.set_state(state9, hist[:, :4].T.copy(), np.full(R, 3, dtype=np.int32))
    b.step(1)
    fin = b.final()
    d = b.d_ray()
    b.close()
    assert np.all(d[2] == 4)
    out = fin[:6].T
    err = np.abs(out - ref) / np.maximum(np.abs(ref), 1e-3)
    err[:, 2] = np.abs(out[:, 2] - ref[:, 2])                          # angles: absolute (radians)
    # golden-section methods included: their comparison sequence is the reference's (rt_exact.h), on all 64 states
    assert err.max() < 1e-12


def test_step_token_call_surface(rb, gpu_fields):
    """op6(i_angle, init_n, i_grad, i_unitv, i_vpos, coef_i, grd, z, step) as the reference calls it (:868)."""
    g = golden("step_methods")
    F = gpu_fields("vert_heterogeneous")
    st, ref = g["st6"][0], g["out6"][0]
    z, grd = rb.FieldSpline(F, "n"), (rb.FieldSpline(F, "dy"), rb.FieldSpline(F, "dx"))
    u = np.array((np.cos(st[2]), np.sin(st[2])))
    fp, fa, fn, fg = rb.op6(st[2], st[3], st[4:6], u, st[0:2], st[6], grd, z, float(g["step"]))
    assert relerr(np.array([fp[0], fp[1], fa, fn, fg[0], fg[1]]), ref, 1e-3) < 1e-12


# ------------------------------------------------------------------ trajectories vs the reference's goldens
@pytest.mark.parametrize("name,scen,m", traj_fixtures())
def test_trajectory_vs_reference(name, scen, m, rb, gpu_fields):
    t = golden("traj_" + name)
    F = gpu_fields(scen)
    x0, y0, th = traj_inputs(t, scen)
    b = rb.Batch(F, m, float(t["step"]), int(t["max_size"]), t["box"], float(t["gamma"]), th, x0, y0, record_stride=1)
    b.run()
    d = b.d_ray(); s = b.rows()
    b.close()
    assert s.shape == (int(t["max_size"]), 6, len(th))
    strided, last = sub_rows(s, d, int(t["stride"]))
    # every ray of every method (north star: 1e-9 on every ray): the field is the reference's bits, op3/4/5/9/10/11 run in its
    # operation order with numpy's own exp / arctan2 / sin / cos restated (<= 3e-17 here); op1/2/6/7/8 in fused forms (~1e-13)
    # (op7 steps in the reference's operation order by default too: it differentiates positions, and a fused position update
    # walked 1.2e-9 .. 8.4e-9 away from the reference on the interface scenario; rtmi_params.reference_order, include/rtmi.h)
    tol = REL
    assert np.array_equal(d[2], t["d_ray"][2])
    assert relerr(strided, t["strided"]) < tol and relerr(last, t["last"]) < tol
    assert relerr(d[:2], t["d_ray"][:2]) < tol
    k = int(np.argmin(d[2])); i = int(d[2, k])
    assert not s[i + 1:, :, k].any()                                   # rows after termination stay zero (Q7)


def test_trazar_show_prints_reference_table(rb, gpu_fields, capsys):
    t = golden("traj_interface_op6_16")
    F = gpu_fields("interface")
    z, grd = rb.FieldSpline(F, "n"), (rb.FieldSpline(F, "dy"), rb.FieldSpline(F, "dx"))
    rb.trazar(rb.op6, z, grd, True, rb.DELTA_S, 91, "1", thetas=t["theta"])
    lines = [l for l in capsys.readouterr().out.splitlines() if l.startswith("Coords:")]
    assert len(lines) == 16 and all("SnellAng:" in l and "InitAng:" in l for l in lines)
    err0 = float(lines[0].split("Err:")[1].split("|")[0])
    assert abs(err0 - t["errors"][0]) < 1e-6


def test_trazar_call_surface_interface(rb, gpu_fields):
    """trazar(selected_func, z, grd, show, step, divisor, user_choice) with the 16-ray cfg1 batch."""
    t = golden("traj_interface_op6_16")
    F = gpu_fields("interface")
    z, grd = rb.FieldSpline(F, "n"), (rb.FieldSpline(F, "dy"), rb.FieldSpline(F, "dx"))
    s_ray, d_ray, compute_times, errors = rb.trazar(rb.op6, z, grd, False, rb.DELTA_S, 91, "1", thetas=t["theta"])
    assert s_ray.shape == (30228, 6, 16) and d_ray.shape == (3, 16) and compute_times.shape == (16,)
    assert np.array_equal(d_ray[2], t["d_ray"][2])
    assert np.abs(errors - t["errors"]).max() < 1e-6                   # degrees; Snell metric (:896-919)
    assert abs(np.mean(errors) - np.mean(t["errors"])) < 1e-7


# ------------------------------------------------------------------ oracle parity at larger, synthetic sizes
@pytest.mark.parametrize("scen,m,R,gam", [("vert_heterogeneous", 6, 4096, 1), ("vert_heterogeneous", 2, 1000, 1),
                                          ("vert_heterogeneous", 3, 1000, 1), ("vert_heterogeneous", 7, 1000, 1),
                                          ("vert_heterogeneous", 8, 1000, 1), ("interface", 6, 777, 1),
                                          ("anisotropy", 11, 130, 3), ("vert_heterogeneous", 9, 130, 1)])
def test_batch_vs_oracle(scen, m, R, gam, rb, gpu_fields, oracle_fields):
    from oracle import rt_oracle as O
    lim = LIMITS[scen]
    th = np.linspace(0.05 if scen == "interface" else 0, np.pi / 2, R)
    max_size = int(np.ceil(80 / rb.DELTA_S) + 1)
    b = rb.Batch(gpu_fields(scen), m, rb.DELTA_S, max_size, lim, gam, th, -2.0, -2.0, record_stride=0)
    b.run()
    d, fin, st = b.d_ray(), b.final(), b.stats()
    b.close()
    o = O.trazar(oracle_fields(scen), m, gam, rb.DELTA_S, max_size, lim, -2.0, -2.0, th, record_stride=0, nthreads=8)
    assert st["ray_steps"] == int(d[2].sum()) and st["live_rays"] == 0
    assert np.array_equal(d[2], o["d_ray"][2])
    if m in (3, 9, 11):       # reference-order methods on a device-built vert field: the oracle's bits
        assert np.array_equal(fin, o["final"]) and np.array_equal(d, o["d_ray"])
    assert relerr(fin, o["final"]) < REL
    assert relerr(d[:2], o["d_ray"][:2]) < REL


def test_fisheye_fan_vs_oracle(rb, gpu_fields, oracle_fields):
    from oracle import rt_oracle as O
    R = 1024
    th = np.linspace(np.pi / 4, 3 * np.pi / 4, R)
    step, max_size, lim = 2 * np.pi / 303, 10 * 304, LIMITS["fisheye"]
    b = rb.Batch(gpu_fields("fisheye"), 6, step, max_size, lim, 1, th, 1.0, 0.0, record_stride=0)
    b.run()
    d, fin = b.d_ray(), b.final()
    b.close()
    o = O.trazar(oracle_fields("fisheye"), 6, 1, step, max_size, lim, 1.0, 0.0, th, record_stride=0, nthreads=8)
    assert np.array_equal(d[2], o["d_ray"][2])
    assert d[2].max() == max_size - 1 and d[2].min() < 1000           # both terminations occur
    assert relerr(fin, o["final"]) < REL


# ------------------------------------------------------------------ launch structure must not change results
def test_one_step_launches_equal_single_launch(rb, gpu_fields):
    F = gpu_fields("vert_heterogeneous")
    th = np.linspace(0, np.pi / 2, 200)
    kw = dict(record_stride=1)
    a = rb.Batch(F, 6, rb.DELTA_S, 400, LIMITS["vert_heterogeneous"], 1, th, -2.0, -2.0, **kw)
    a.run()
    b = rb.Batch(F, 6, rb.DELTA_S, 400, LIMITS["vert_heterogeneous"], 1, th, -2.0, -2.0, **kw)
    for _ in range(150):
        b.step(1)
    b.step(7)
    b.step(1000)
    assert b.stats()["launches"] == 152
    assert np.array_equal(a.d_ray(), b.d_ray()) and np.array_equal(a.final(), b.final())
    assert np.array_equal(a.rows(), b.rows())
    assert np.all(a.d_ray()[2] == 399)                                  # max_size exhausted: last row max_size-1
    b.reset(); b.run()
    assert np.array_equal(a.rows(), b.rows())                           # reset re-runs from row 0
    a.close(); b.close()


@pytest.mark.parametrize("dtype", ["f64", "f32"])
def test_launch_cuts_across_unit_vector_refresh(dtype, rb, gpu_fields):
    """op2/op6 carry the unit tangent as state and rotate it (recomputed from the angle every 1024th row of a ray, fp64):
    a trajectory must not depend on where launches cut it, including right at and around the refresh rows."""
    F = gpu_fields("vert_heterogeneous", rb.F64 if dtype == "f64" else rb.F32)
    th = np.linspace(np.pi / 4 - 0.02, np.pi / 4 + 0.02, 192)           # rays that stay inside past the second refresh row
    kw = dict(record_stride=1)
    a = rb.Batch(F, 6, rb.DELTA_S, 2200, LIMITS["vert_heterogeneous"], 1, th, -2.0, -2.0, **kw)
    a.run()
    b = rb.Batch(F, 6, rb.DELTA_S, 2200, LIMITS["vert_heterogeneous"], 1, th, -2.0, -2.0, **kw)
    for n in (1000, 23, 1, 1, 3, 1019, 1, 2, 5000):                     # cuts at rows 1000, 1023, 1024, 1025, 1028, 2047, 2048, 2050
        b.step(n)
    assert a.d_ray()[2].min() > 2060
    assert np.array_equal(a.d_ray(), b.d_ray()) and np.array_equal(a.final(), b.final())
    assert np.array_equal(a.rows(), b.rows())
    a.close(); b.close()


@pytest.mark.parametrize("m", [6, 7])
def test_sharding_is_bit_identical(m, rb, gpu_fields):
    """rays are independent, so any partition gives the same bits."""
    F = gpu_fields("vert_heterogeneous")
    R = 1000
    th = np.linspace(0, np.pi / 2, R)
    ms = int(np.ceil(80 / rb.DELTA_S) + 1)
    whole = rb.Batch(F, m, rb.DELTA_S, ms, LIMITS["vert_heterogeneous"], 1, th, -2.0, -2.0, record_stride=32)
    whole.run()
    W, Wd, Wf = whole.rows(), whole.d_ray(), whole.final()
    whole.close()
    for lo, hi in ((0, 333), (333, 334), (334, 1000)):
        p = rb.Batch(F, m, rb.DELTA_S, ms, LIMITS["vert_heterogeneous"], 1, th[lo:hi], -2.0, -2.0, record_stride=32)
        p.run()
        assert np.array_equal(p.rows(), W[:, :, lo:hi]) and np.array_equal(p.d_ray(), Wd[:, lo:hi])
        assert np.array_equal(p.final(), Wf[:, lo:hi])
        p.close()


@pytest.mark.parametrize("scen,m,dtype,slice_steps,stride", [("vert_heterogeneous", 6, "f64", 0, 1), ("fisheye", 6, "f64", 100, 1),
                                                         ("vert_heterogeneous", 7, "f64", 37, 16), ("vert_heterogeneous", 9, "f64", 64, 1),
                                                         ("vert_heterogeneous", 6, "f32", 300, 1), ("interface", 2, "f64", 1000, 0)])
def test_sliced_bundles_are_bit_identical(scen, m, dtype, slice_steps, stride, rb, gpu_fields, oracle_fields):
    """RTMI_LAUNCH_SLICED (persistent blocks, 256-ray bundles advanced in time slices through a ticket counter) vs one lane
    per ray to completion: same rows, same final state, same step counts; more bundles than resident blocks would need
    a large batch, so the ordering of a bundle's slices is exercised with short slices instead."""
    R = 5000 if m != 9 else 600
    lim = LIMITS[scen]
    if scen == "fisheye":
        th = np.linspace(np.pi / 4, 3 * np.pi / 4, R); x0, y0 = 1.0, 0.0
        step, ms = 2 * np.pi / 303, 3040
    else:
        th = np.linspace(0.06, np.pi / 2, R); x0, y0 = -2.0, -2.0
        step, ms = rb.DELTA_S, int(np.ceil(80 / rb.DELTA_S) + 1)
    F = gpu_fields(scen, rb.F64 if dtype == "f64" else rb.F32)
    a = rb.Batch(F, m, step, ms, lim, 1, th, x0, y0, record_stride=stride, launch_mode="plain")
    a.run()
    b = rb.Batch(F, m, step, ms, lim, 1, th, x0, y0, record_stride=stride, launch_mode="sliced", slice_steps=slice_steps)
    b.run()
    sa, sb = a.stats(), b.stats()
    assert sa["ray_steps"] == sb["ray_steps"] == int(a.d_ray()[2].sum()) and sb["live_rays"] == 0 and sb["launches"] == 1
    assert np.array_equal(a.d_ray(), b.d_ray()) and np.array_equal(a.final(), b.final())
    if stride:
        assert np.array_equal(a.rows(), b.rows())
    b.reset(); b.step(123); b.run()            # part of the way with the plain kernel, then the sliced launch
    assert np.array_equal(a.d_ray(), b.d_ray()) and np.array_equal(a.final(), b.final())
    if stride:
        assert np.array_equal(a.rows(), b.rows())
    if m == 9:                                  # and against the oracle directly: the reference-order method's bits
        from oracle import rt_oracle as O
        o = O.trazar(oracle_fields(scen), m, 1, step, ms, lim, x0, y0, th, record_stride=stride, nthreads=8)
        assert np.array_equal(b.final(), o["final"]) and np.array_equal(b.rows(), o["s_ray"])
    a.close(); b.close()


def test_trazar_auto_launch_mode_is_the_plain_launch_bit_for_bit(rb, gpu_fields):
    """The reference call surface picks the time-sliced schedule by itself for large batches; the return values must not notice."""
    F = gpu_fields("vert_heterogeneous")
    z, grd = rb.FieldSpline(F, "n"), (rb.FieldSpline(F, "dy"), rb.FieldSpline(F, "dx"))
    th = np.linspace(0.0, np.pi / 2, 70000)
    outs = [rb.trazar(rb.op6, z, grd, False, rb.DELTA_S, 91, "3", thetas=th, record=16, launch_mode=lm) for lm in ("auto", "plain")]
    assert np.array_equal(outs[0][0], outs[1][0]) and np.array_equal(outs[0][1], outs[1][1])


def test_sliced_bundles_randomised(rb, gpu_fields):
    """RTMI_LAUNCH_SLICED vs the plain launch over seeded random shapes: ray counts from 1 to 300 000, slices from 1 to 5 000
    steps, methods of every class, record strides 0 / 1 / 5, sorted and shuffled fans, both precisions."""
    rng = np.random.default_rng(2026)
    for case in range(18):
        scen = ["vert_heterogeneous", "fisheye", "interface"][case % 3]
        m = int(rng.choice([1, 2, 6, 7, 8, 9, 3]))
        dt = rb.F32 if (case % 7 == 3 and m in (1, 2, 6, 7, 8)) else rb.F64
        R = int(rng.choice([1, 63, 257, 1000, 5000, 40000, 300000])) if m not in (9, 3) else int(rng.choice([1, 300, 3000]))
        sl = int(rng.choice([1, 7, 50, 128, 333, 1024, 5000]))
        stride = int(rng.choice([0, 1, 5]))
        if scen == "fisheye":
            th = np.sort(rng.uniform(np.pi / 4, 3 * np.pi / 4, R)); x0, y0, step, ms = 1.0, 0.0, 2 * np.pi / 303, 3040
        else:
            th = np.sort(rng.uniform(0.06, np.pi / 2, R)); x0, y0, step = -2.0, -2.0, rb.DELTA_S
            ms = 4000 if scen == "vert_heterogeneous" else 6000
        if rng.random() < 0.3:
            th = rng.permutation(th)
        kw = dict(record_stride=stride, rec_rows=min(ms, 1200) if stride == 1 and R > 20000 else 0)
        F = gpu_fields(scen, dt)
        a = rb.Batch(F, m, step, ms, LIMITS[scen], 1, th, x0, y0, launch_mode="plain", **kw)
        a.run()
        b = rb.Batch(F, m, step, ms, LIMITS[scen], 1, th, x0, y0, launch_mode="sliced", slice_steps=sl, **kw)
        b.run()
        tag = f"case {case}: {scen} op{m} R={R} slice={sl} stride={stride}"
        assert b.stats()["live_rays"] == 0, tag
        assert np.array_equal(a.d_ray(), b.d_ray()) and np.array_equal(a.final(), b.final()), tag
        if stride and R <= 40000:
            assert np.array_equal(a.rows(), b.rows()), tag
        a.close(); b.close()


def test_sliced_bundles_more_bundles_than_resident_blocks(rb, gpu_fields):
    """RTMI_LAUNCH_SLICED with 1 954 bundles for at most 1 024 resident blocks and short slices: tickets of one bundle are drawn
    by different blocks while its previous slice may still be running elsewhere (the per-bundle ordering), bundles die at
    very different rows (the fisheye fan); the final state must still be the lane kernel's bits."""
    R = 500_000
    lim = LIMITS["fisheye"]
    th = np.linspace(np.pi / 4, 3 * np.pi / 4, R)
    F = gpu_fields("fisheye")
    a = rb.Batch(F, 6, 2 * np.pi / 303, 3040, lim, 1, th, 1.0, 0.0, record_stride=0, launch_mode="plain")
    a.run()
    b = rb.Batch(F, 6, 2 * np.pi / 303, 3040, lim, 1, th, 1.0, 0.0, record_stride=0, launch_mode="sliced", slice_steps=96)
    b.run()
    assert b.stats()["live_rays"] == 0 and a.stats()["ray_steps"] == b.stats()["ray_steps"]
    assert np.array_equal(a.d_ray(), b.d_ray()) and np.array_equal(a.final(), b.final())
    a.close(); b.close()


def test_sliced_long_slices_do_not_trip_the_wait_bound(rb, gpu_fields):
    """The time-sliced kernel bounds its waits by wall-clock time WITHOUT PROGRESS, scaled to the longest possible slice --
    a block that waits for the last live bundle is not a stall.  (1) op11 (74 cost evaluations per step) with slice_steps
    2^20: every bundle runs to its end in its first slice, idle blocks wait for the slowest.  (2) one long-lived bundle
    among 1 171 short ones, more bundles than resident blocks, slice longer than max_size (nothing is ever pushed back)."""
    lim = LIMITS["anisotropy"]
    th = np.linspace(0, np.pi / 2, 20000)
    ms = int(np.ceil(80 / rb.DELTA_S) + 1)
    a = rb.Batch(gpu_fields("anisotropy"), 11, rb.DELTA_S, ms, lim, 3, th, -2.0, -2.0, record_stride=0, launch_mode="plain")
    a.run()
    b = rb.Batch(gpu_fields("anisotropy"), 11, rb.DELTA_S, ms, lim, 3, th, -2.0, -2.0, record_stride=0, launch_mode="sliced",
                 slice_steps=1 << 20)
    b.run()                                                  # raises RtmiError(RTMI_ERR_STATE) if a wait was abandoned
    assert b.stats()["live_rays"] == 0 and np.array_equal(a.final(), b.final()) and np.array_equal(a.d_ray(), b.d_ray())
    a.close(); b.close()
    R = 300_000
    th = np.full(R, np.pi / 4 + 0.01)                        # leaves the fisheye box after a few hundred rows ...
    th[150_000:150_256] = np.pi / 2                          # ... except one bundle on the closed circle: all 3 039 rows
    F = gpu_fields("fisheye")
    a = rb.Batch(F, 6, 2 * np.pi / 303, 3040, LIMITS["fisheye"], 1, th, 1.0, 0.0, record_stride=0, launch_mode="plain")
    a.run()
    b = rb.Batch(F, 6, 2 * np.pi / 303, 3040, LIMITS["fisheye"], 1, th, 1.0, 0.0, record_stride=0, launch_mode="sliced",
                 slice_steps=5000)
    b.run()
    d = b.d_ray()
    assert d[2].max() == 3039 and np.median(d[2]) < 1000
    assert np.array_equal(a.final(), b.final()) and np.array_equal(a.d_ray(), d)
    a.close(); b.close()


@pytest.mark.parametrize("scen,m,dtype", [("vert_heterogeneous", 6, "f64"), ("fisheye", 2, "f64"), ("vert_heterogeneous", 7, "f64"),
                                          ("vert_heterogeneous", 9, "f64"), ("vert_heterogeneous", 6, "f32")])
def test_checkpoint_resume_is_bit_identical(scen, m, dtype, rb, gpu_fields):
    """rtmi_batch_get_state -> rtmi_batch_restore_state on a fresh batch continues a run bit for bit: the whole ray state
    travels, including op7's position history and the unit tangent fp64 op2/op6 carry by rotation (hist4 rows 0-1)."""
    R = 700
    lim = LIMITS[scen]
    if scen == "fisheye":
        th, x0, y0, step, ms = np.linspace(np.pi / 4, 3 * np.pi / 4, R), 1.0, 0.0, 2 * np.pi / 303, 3040
    else:
        th, x0, y0, step, ms = np.linspace(0.06, np.pi / 2, R), -2.0, -2.0, rb.DELTA_S, 30228
    F = gpu_fields(scen, rb.F64 if dtype == "f64" else rb.F32)
    a = rb.Batch(F, m, step, ms, lim, 1, th, x0, y0, record_stride=1, rec_rows=3100)
    a.run()
    b = rb.Batch(F, m, step, ms, lim, 1, th, x0, y0, record_stride=1, rec_rows=3100)
    b.step(777)
    st, hist, istep, live = b.get_state()
    if m in (2, 6) and dtype == "f64":
        assert np.allclose(hist[0] ** 2 + hist[1] ** 2, 1.0, atol=1e-12) and not hist[2:].any()
    c = rb.Batch(F, m, step, ms, lim, 1, th, x0, y0, record_stride=1, rec_rows=3100)
    c.restore_state(st, hist, istep, live)
    c.run()
    assert np.array_equal(c.d_ray(), a.d_ray()) and np.array_equal(c.final(), a.final())
    assert np.array_equal(c.rows(778, 2300), a.rows(778, 2300))      # the rows written after the resume
    at = live == 1                                            # rtmi_read_final's momenta are the last recorded row's
    assert at.any() and np.array_equal(b.final()[6:8][:, at], b.rows(777, 1)[0, 2:4][:, at])
    a.close(); b.close(); c.close()


@pytest.mark.parametrize("scen,m,refill_min", [("vert_heterogeneous", 6, 0), ("vert_heterogeneous", 7, 1),
                                               ("interface", 6, 48), ("vert_heterogeneous", 9, 16)])
def test_lane_refill_is_bit_identical(scen, m, refill_min, rb, gpu_fields, oracle_fields):
    """RTMI_LAUNCH_REFILL (persistent waves, ballot/mbcnt compaction of terminated lanes) vs one lane per ray, on a
    batch whose rays are in random order (neighbouring lanes terminate hundreds of steps apart)."""
    rng = np.random.default_rng(5)
    R = 3000 if m != 9 else 300
    lim = LIMITS[scen]
    th = rng.permutation(np.linspace(0.06, np.pi / 2, R))
    ms = int(np.ceil(80 / rb.DELTA_S) + 1)
    a = rb.Batch(gpu_fields(scen), m, rb.DELTA_S, ms, lim, 1, th, -2.0, -2.0, record_stride=16, launch_mode="plain")
    a.run()
    b = rb.Batch(gpu_fields(scen), m, rb.DELTA_S, ms, lim, 1, th, -2.0, -2.0, record_stride=16, launch_mode="refill",
                 refill_min=refill_min)
    b.step(100)                                # part of the way with the plain kernel, then drain the queue
    b.run()
    sa, sb = a.stats(), b.stats()
    assert sa["ray_steps"] == sb["ray_steps"] == int(a.d_ray()[2].sum()) and sb["live_rays"] == 0
    assert np.array_equal(a.d_ray(), b.d_ray()) and np.array_equal(a.final(), b.final())
    assert np.array_equal(a.rows(), b.rows())
    b.reset(); b.run()                         # and from row 0
    assert np.array_equal(a.final(), b.final()) and np.array_equal(a.rows(), b.rows())
    # the refill kernel against the oracle directly (not only against the other kernel)
    from oracle import rt_oracle as O
    o = O.trazar(oracle_fields(scen), m, 1, rb.DELTA_S, ms, lim, -2.0, -2.0, th, record_stride=16, nthreads=8)
    assert np.array_equal(b.d_ray()[2], o["d_ray"][2])
    if m == 9:                                  # reference-order method: the oracle's bits
        assert np.array_equal(b.final(), o["final"]) and np.array_equal(b.rows(), o["s_ray"])
    else:
        assert relerr(b.final(), o["final"]) < REL and relerr(b.rows(), o["s_ray"]) < REL
    a.close(); b.close()


@pytest.mark.parametrize("scen,m", [("vert_heterogeneous", 6), ("interface", 6), ("fisheye", 6), ("vert_heterogeneous", 7)])
def test_uniform_basis_vs_exact_basis(scen, m, rb, gpu_fields, oracle_fields):
    """rtmi_params.exact_basis is accepted and has no effect any more: the fast-form methods evaluate one polynomial per grid
    cell converted from FITPACK's splines on the true knots of EVERY cell (rt_polytab.h; rounds 1-2 used uniform-knot weights
    in interior cells, <= 4e-14 off, and this flag selected fpbspl on the true knots).  Both settings: the same bits, and
    within the 1e-9 tolerance of the oracle (measured ~1e-14 on vert_heterogeneous)."""
    from oracle import rt_oracle as O
    R = 512
    lim = LIMITS[scen]
    if scen == "fisheye":
        th, x0, y0, step, ms = np.linspace(np.pi / 4, 3 * np.pi / 4, R), 1.0, 0.0, 2 * np.pi / 303, 3040
    else:
        th, x0, y0, step, ms = np.linspace(0.06, np.pi / 2, R), -2.0, -2.0, rb.DELTA_S, 30228
    res = []
    for exact in (0, 1):
        b = rb.Batch(gpu_fields(scen), m, step, ms, lim, 1, th, x0, y0, record_stride=0, exact_basis=exact)
        b.run()
        res.append((b.d_ray(), b.final()))
        b.close()
    assert np.array_equal(res[0][0][2], res[1][0][2])
    gap = relerr(res[0][1], res[1][1])
    o = O.trazar(oracle_fields(scen), m, 1, step, ms, lim, x0, y0, th, record_stride=0, nthreads=8)
    e_fast, e_exact = relerr(res[0][1], o["final"]), relerr(res[1][1], o["final"])
    print(f"{scen} op{m}: fast vs exact {gap:.2e}; vs oracle: fast {e_fast:.2e}, exact {e_exact:.2e}")
    assert gap == 0.0 and e_fast < REL and e_exact < REL


@pytest.mark.parametrize("scen,m,shuffle,mode", [("vert_heterogeneous", 6, False, "plain"), ("vert_heterogeneous", 6, True, "plain"),
                                                 ("interface", 6, False, "plain"), ("fisheye", 6, False, "plain"),
                                                 ("vert_heterogeneous", 7, True, "refill"), ("anisotropy", 11, False, "plain")])
def test_lds_tile_equals_global_gather(scen, m, shuffle, mode, rb, gpu_fields):
    """field_path 2 -- the wave-shared lookup: for the fast-form methods the cell's polynomial through the scalar cache
    (wave-uniform cell, per-lane loads for lanes elsewhere), for the reference-order methods the wave-private LDS tile
    (re-staged as the wave moves; lanes outside fall back to global loads) -- must give the same bits as field_path 1
    (every lane reads for itself): coherent fans, a shuffled batch, the refill kernel, rays that run to the grid's edge."""
    R = 2000 if m != 11 else 200
    lim = LIMITS[scen]
    if scen == "fisheye":
        th, x0, y0, step, ms = np.linspace(np.pi / 4, 3 * np.pi / 4, R), 1.0, 0.0, 2 * np.pi / 303, 3040
    else:
        th, x0, y0, step, ms = np.linspace(0.06, np.pi / 2, R), -2.0, -2.0, rb.DELTA_S, 30228
    if shuffle:
        th = np.random.default_rng(9).permutation(th)
    if scen == "vert_heterogeneous" and not shuffle:
        lim = (-4.9, 7.9, -5.4, 3.9)          # let rays run into the not-a-knot end cells of the grid
    out = []
    for path in (2, 1):
        b = rb.Batch(gpu_fields(scen), m, step, ms, lim, 3 if scen == "anisotropy" else 1, th, x0, y0, record_stride=8,
                     field_path=path, launch_mode=mode)
        b.run()
        out.append((b.d_ray(), b.final(), b.rows(), b.stats()["lds_bytes"]))
        b.close()
    if m == 11:
        assert out[0][3] > 16384 and out[1][3] <= 4096  # the tile variant really carries the LDS tile (the other: glibc's 3.5 KB sin/cos table)
    for u, v in zip(out[0][:3], out[1][:3]):
        assert np.array_equal(u, v)


@pytest.mark.parametrize("scen,m,shuffle,mode", [("vert_heterogeneous", 7, False, "plain"), ("vert_heterogeneous", 3, False, "plain"),
                                                 ("vert_heterogeneous", 6, False, "plain"), ("vert_heterogeneous", 9, False, "sliced"),
                                                 ("vert_heterogeneous", 4, True, "plain"), ("interface", 7, False, "plain"),
                                                 ("interface", 5, False, "plain"), ("fisheye", 3, False, "plain"),
                                                 ("vert_heterogeneous", 7, False, "refill"), ("anisotropy", 11, False, "plain"),
                                                 ("anisotropy", 10, False, "sliced")])
def test_uniform_window_equals_per_lane_lookup(scen, m, shuffle, mode, rb, gpu_fields, oracle_fields):
    """field_path 3 -- the reference-order lookup of a wave whose live lanes share one grid cell reads the 4 x 4 (and 2 x 2) coefficient
    window, the cell's knots, the seven knot differences of each axis and their reciprocals ONCE through the scalar cache
    (rt::ex::lookup_uniform; auto's choice from 131 072 rays on) -- must give the same bits as field_path 1 (every lane gathers its
    own window), and both the oracle's: coherent fans (vert: one cell per wave on most steps; interface and fisheye: waves that
    straddle cells take the per-lane path on those steps), a shuffled batch (never uniform), the refill and sliced schedules, rays
    that run into the not-a-knot end cells and out of the grid (FITPACK's clamp), op6 with reference_order."""
    from oracle import rt_oracle as O
    R = 2000 if m not in (5, 9, 10, 11) else 256
    lim = LIMITS[scen]
    if scen == "fisheye":
        th, x0, y0, step, ms = np.linspace(np.pi / 4, 3 * np.pi / 4, R), 1.0, 0.0, 2 * np.pi / 303, 3040
    else:
        th, x0, y0, step, ms = np.linspace(0.06, np.pi / 2, R), -2.0, -2.0, rb.DELTA_S, 30228
    if shuffle:
        th = np.random.default_rng(9).permutation(th)
    if scen == "vert_heterogeneous" and not shuffle and m in (7, 3):
        lim = (-5.2, 8.2, -5.7, 4.2)          # through the not-a-knot end cells and past the grid's rim (the grid ends 3 units outside the default box)
    gam = 3 if scen == "anisotropy" else 1
    out = []
    for path in (3, 1):
        b = rb.Batch(gpu_fields(scen), m, step, ms, lim, gam, th, x0, y0, record_stride=8, field_path=path, launch_mode=mode,
                     reference_order=(m == 6))
        b.run()
        out.append((b.d_ray(), b.final(), b.rows()))
        b.close()
    for u, v in zip(out[0], out[1]):
        assert np.array_equal(u, v)
    o = O.trazar(oracle_fields(scen), m, gam, step, ms, lim, x0, y0, th, record_stride=8, nthreads=16)
    assert np.array_equal(out[0][0], o["d_ray"]) and np.array_equal(out[0][1], o["final"]) and np.array_equal(out[0][2], o["s_ray"])


def test_record_strides_and_edges(rb, gpu_fields):
    F = gpu_fields("vert_heterogeneous")
    lim = LIMITS["vert_heterogeneous"]
    th = np.array([0.3])
    full = rb.Batch(F, 6, rb.DELTA_S, 3000, lim, 1, th, -2.0, -2.0, record_stride=1)
    full.run()
    S, nr = full.rows(want_n_ray=True)
    last = int(full.d_ray()[2, 0])
    for stride in (2, 7, 64):
        b = rb.Batch(F, 6, rb.DELTA_S, 3000, lim, 1, th, -2.0, -2.0, record_stride=stride)
        b.run()
        s2, n2 = b.rows(want_n_ray=True)
        assert s2.shape[0] == (3000 + stride - 1) // stride
        assert np.array_equal(s2, S[::stride]) and np.array_equal(n2, nr[::stride])
        b.close()
    # traveltime recurrence (Q6) holds on the recorded rows
    dist = np.hypot(np.diff(S[:last + 1, 0, 0]), np.diff(S[:last + 1, 1, 0]))
    T = np.concatenate(([0], np.cumsum(dist * (nr[:last, 0] + nr[1:last + 1, 0]) / 2)))
    assert np.abs(T - S[:last + 1, 4, 0]).max() < 1e-13
    full.close()
    # a ray launched outside the box still takes one step and stores it (Q7)
    b = rb.Batch(F, 6, rb.DELTA_S, 100, lim, 1, [0.1, 0.1], [-2.5, -2.0], [-2.0, -2.0], record_stride=1)
    b.run()
    assert list(b.d_ray()[2]) == [1, 99]
    b.close()
    from raytracing_amd._lib import RtmiError
    with pytest.raises(ValueError):
        rb.Batch(F, 12, rb.DELTA_S, 100, lim, 1, th, -2.0, -2.0)
    with pytest.raises(RtmiError, match="op7"):
        rb.Batch(F, 7, rb.DELTA_S, 3, lim, 1, th, -2.0, -2.0)


# ------------------------------------------------------------------ full-size properties (BASELINE configs)
def test_cfg2_vert_65536_properties(rb, gpu_fields, oracle_fields):
    """cfg2: vert_heterogeneous, 65 536 rays fp64.  p_x = n cos(theta) is conserved,
    step counts are checked against the oracle on a 1/256 subsample, and totals on the device counter."""
    from oracle import rt_oracle as O
    R = 65536
    th = np.linspace(0, np.pi / 2, R)
    lim = LIMITS["vert_heterogeneous"]
    ms = int(np.ceil(80 / rb.DELTA_S) + 1)
    b = rb.Batch(gpu_fields("vert_heterogeneous"), 6, rb.DELTA_S, ms, lim, 1, th, -2.0, -2.0, record_stride=0)
    b.run()
    d, fin, st = b.d_ray(), b.final(), b.stats()
    b.close()
    assert st["ray_steps"] == int(d[2].sum())
    assert 1006 <= d[2].min() and d[2].max() <= 2960               # 2 953 at this fan density
    px0 = 0.07142864686293911 * np.cos(th)
    assert np.max(np.abs(fin[6] - px0)[1:-1] / px0[1:-1]) < 5e-4          # CV threshold scale (:1310)
    sub = slice(0, R, 256)
    o = O.trazar(oracle_fields("vert_heterogeneous"), 6, 1, rb.DELTA_S, ms, lim, -2.0, -2.0, th[sub], record_stride=0,
                 nthreads=8)
    assert np.array_equal(d[2][sub], o["d_ray"][2]) and relerr(fin[:, sub], o["final"]) < REL


def test_cfg2_vert_65536_full_record_rows(rb, gpu_fields, oracle_fields):
    """cfg2 with the reference's full record -- s_ray[3072][6][65 536], the few-waves kernel's RECORDING build (k_advance_lat, what
    a 65 536-ray batch runs; rtmi_stats says so: 200+ registers) -- row for row against the oracle on every 64th ray (1 024 rays x
    every row they have): step counts equal, every recorded quantity within 1e-9, rows past a ray's last one zero (:802)."""
    from oracle import rt_oracle as O
    R = 65536
    th = np.linspace(0, np.pi / 2, R)
    lim = LIMITS["vert_heterogeneous"]
    ms = int(np.ceil(80 / rb.DELTA_S) + 1)
    rows = 3072
    b = rb.Batch(gpu_fields("vert_heterogeneous"), 6, rb.DELTA_S, ms, lim, 1, th, -2.0, -2.0, record_stride=1, rec_rows=rows, keep_n_ray=False)
    b.run()
    d, fin, st = b.d_ray(), b.final(), b.stats()
    sub = slice(0, R, 64)
    got = b.device_tensors()["s_ray"][:, :, sub].cpu().numpy()
    b.close()
    assert st["vgprs"] > 128 and st["launch_mode_used"] == "plain"          # the few-waves build (two waves per SIMD: 165 VGPRs), not k_advance's
    o = O.trazar(oracle_fields("vert_heterogeneous"), 6, 1, rb.DELTA_S, ms, lim, -2.0, -2.0, th[sub], record_stride=1, rec_rows=rows, nthreads=16)
    assert np.array_equal(d[2][sub], o["d_ray"][2])
    assert got.shape == o["s_ray"].shape
    err = relerr(got, o["s_ray"])
    print(f"cfg2 full record, {got.shape[2]} rays x {got.shape[0]} rows vs the oracle: {err:.2e}; final state {relerr(fin[:, sub], o['final']):.2e}")
    assert err < REL and relerr(fin[:, sub], o["final"]) < REL and relerr(d[:2, sub], o["d_ray"][:2]) < REL
    last = o["d_ray"][2].astype(int)
    for k in range(0, got.shape[2], 16):
        assert not got[last[k] + 1:, :, k].any()


def test_fp32_path_tracks_fp64(rb, gpu_fields):
    """cfg4 runs an fp32 field and fp32 step arithmetic on fp64 accumulators (position, angle, arclengths, traveltime);
    the reference is fp64-only, so the tolerance is measured, not inherited: same step count on (nearly) every ray
    -- a ray may cross the box edge one row earlier or later -- and end points within 2e-5 of the fp64 trajectory."""
    R = 512
    th = np.linspace(0, np.pi / 2, R)
    lim = LIMITS["vert_heterogeneous"]
    ms = int(np.ceil(80 / rb.DELTA_S) + 1)
    a = rb.Batch(gpu_fields("vert_heterogeneous", 0), 6, rb.DELTA_S, ms, lim, 1, th, -2.0, -2.0, record_stride=0)
    b = rb.Batch(gpu_fields("vert_heterogeneous", 1), 6, rb.DELTA_S, ms, lim, 1, th, -2.0, -2.0, record_stride=0)
    a.run(); b.run()
    fa, fb, da, db = a.final(), b.final(), a.d_ray(), b.d_ray()
    a.close(); b.close()
    assert np.max(np.abs(da[2] - db[2])) <= 1
    same = da[2] == db[2]
    err = np.abs(fa[:2] - fb[:2])[:, same].max()
    print(f"fp32 vs fp64: same step count on {same.sum()}/{R} rays, end-point max abs error {err:.3e}, "
          f"traveltime {np.abs(fa[8] - fb[8])[same].max():.3e}")
    assert same.mean() > 0.98 and err < 2e-5


# ------------------------------------------------------------------ on-device validation metrics
def test_device_metrics_match_reference(rb, gpu_fields):
    """rtmi_metric vs the reference's own numbers (goldens) and vs the host restatements on read-back rows."""
    # interface: exit-angle errors (:896-919)
    t = golden("traj_interface_op6_16")
    b = rb.Batch(gpu_fields("interface"), 6, float(t["step"]), int(t["max_size"]), t["box"], 1, t["theta"], t["pos_x"],
                 -2.0, record_stride=1)
    b.run()
    err = b.metric("snell")
    assert np.abs(err - t["errors"]).max() < 1e-6
    assert np.abs(err - rb.snell_errors(b.rows(), b.d_ray(), t["theta"])).max() < 1e-9
    b.close()
    # vert: p_x coefficient of variation (:1354-1360), reference value in the fixture
    t = golden("traj_vert_op6")
    b = rb.Batch(gpu_fields("vert_heterogeneous"), 6, float(t["step"]), int(t["max_size"]), t["box"], 1, t["theta"], -2.0,
                 -2.0, record_stride=1)
    b.run()
    cv = b.metric("px_cv")
    assert abs(np.mean(cv[1:-1]) - float(t["cv_mean"])) < 1e-9
    assert abs(np.mean(cv[1:-1]) - rb.moment_cv(b.rows(), 31)) < 1e-10
    b.close()
    # fisheye: closure error after N turns (:956), reference value 3.0408 %
    t = golden("traj_fisheye_op6_div304")
    b = rb.Batch(gpu_fields("fisheye"), 6, float(t["step"]), int(t["max_size"]), t["box"], 1, t["theta"], 1.0, 0.0,
                 record_stride=0)
    b.run()
    assert abs(b.metric("closure")[0] - float(t["closure_pct"])) < 1e-9
    b.close()
    from raytracing_amd._lib import RtmiError
    b = rb.Batch(gpu_fields("vert_heterogeneous"), 6, rb.DELTA_S, 500, LIMITS["vert_heterogeneous"], 1, [0.3], -2.0, -2.0,
                 record_stride=4)
    b.run()
    with pytest.raises(RtmiError, match="full trajectory"):
        b.metric("snell")
    b.close()


def test_cfg3_fisheye_1m_properties(rb, gpu_fields):
    """cfg3: fisheye, 1 048 576 rays fp64, calibrated op6 step.  Every launch direction theta from (1,0) rides the
    circle through (1,0) and (-1,0) centred at (0, cot theta) (Maxwell fisheye); rays within |theta-pi/2| < 0.38
    stay inside the +-1.5 box for all 3 039 steps and must still be on their circle, the outer ones must leave
    the box; the central ray reproduces the reference's closure error (3.0408 %)."""
    R = 1 << 20
    th = np.linspace(np.pi / 4, 3 * np.pi / 4, R)
    step, ms = 2 * np.pi / 303, 10 * 304
    b = rb.Batch(gpu_fields("fisheye"), 6, step, ms, LIMITS["fisheye"], 1, th, 1.0, 0.0, record_stride=0)
    b.run()
    d, st, cl, fin = b.d_ray(), b.stats(), b.metric("closure"), b.final()
    b.close()
    assert st["ray_steps"] == int(d[2].sum())
    stay = np.abs(th - np.pi / 2) < 0.38
    leave = np.abs(th - np.pi / 2) > 0.41
    assert np.all(d[2][stay] == ms - 1) and np.all(d[2][leave] < ms - 1)
    mid = R // 2
    assert abs(cl[mid] - 3.0408) < 0.01
    yc = 1.0 / np.tan(th[stay])
    radial = np.abs(np.hypot(fin[0][stay], fin[1][stay] - yc) - np.sqrt(1 + yc * yc))
    print(f"fisheye 1M: max radial deviation from the analytic circle after 10 turns: {radial.max():.2e}")
    assert radial.max() < 5e-3


# ------------------------------------------------------------------ DELTA_S calibration sweep
@pytest.mark.parametrize("scen", ["interface", "fisheye", "vert_heterogeneous"])
def test_delta_s_sweep_vs_reference(scen, rb, gpu_fields):
    """search_delta_sweep (device traces + device metrics per candidate) against the reference's search_delta
    on every 10th candidate of its calibration grid (fixtures made by oracle/gen_golden.py --only sweep)."""
    g = golden(f"sweep_{scen}_op6")
    choice = {"interface": "1", "fisheye": "2", "vert_heterogeneous": "3"}[scen]
    div, opt = rb.delta_s_candidates(choice)
    assert np.array_equal(div, g["all_divisors"]) and np.array_equal(opt, g["all_options"])
    sel = g["sel"]
    res = rb.search_delta_sweep(rb.op6, gpu_fields(scen), None, opt[sel], div[sel], choice)           # one batch
    seq = rb.search_delta_sweep(rb.op6, gpu_fields(scen), None, opt[sel], div[sel], choice, batched=False)
    assert np.array_equal(np.array(res), np.array(seq))       # candidate x ray batch == one batch per candidate
    ref = g["results"]
    if scen == "interface":
        assert np.abs(np.array(res) - ref).max() < 1e-6           # degrees (mean, max)
    else:
        assert np.abs(np.array(res) - ref[:, 0]).max() < 1e-8     # closure % / mean p_x CV %
    # the selection rule runs on whatever the sweep returns (thresholds: 0.2 deg & 0.8 deg / 5 % / 0.05 %)
    pick = rb.find_divisor(res, div[sel], choice)
    ref_pick = rb.find_divisor([tuple(r) for r in ref] if scen == "interface" else list(ref[:, 0]), div[sel], choice)
    assert pick == ref_pick


# ------------------------------------------------------------------ isochrone points
def test_isochrones_vs_scipy_pchip(rb, gpu_fields):
    """rtmi_isochrones (device PCHIP per ray) against scipy's PchipInterpolator applied to the reference's own
    trajectories exactly as RT_bench.py:987-1003 does (fixture isochrones_vert_op6)."""
    g = golden("isochrones_vert_op6")
    b = rb.Batch(gpu_fields("vert_heterogeneous"), 6, float(g["step"]), int(g["max_size"]), g["box"], 1, g["theta"], -2.0,
                 -2.0, record_stride=1)
    b.run()
    pts = b.isochrones(g["times"])
    b.close()
    ref = g["points"]
    assert pts.shape == ref.shape == (11, 3, 31)
    assert np.array_equal(np.isnan(pts), np.isnan(ref))          # same rays reach each traveltime
    ok = ~np.isnan(ref)
    assert ok.sum() > 300 and np.abs(pts[ok] - ref[ok]).max() < 1e-10
    # rays are perpendicular to wavefronts in an isotropic medium (:1016-1039): check one isochrone's tangent
    it = 3
    sel = ~np.isnan(pts[it, 0])
    x, y, th = pts[it, 0, sel], pts[it, 1, sel], pts[it, 2, sel]
    tx, ty = np.gradient(x), np.gradient(y)                       # wavefront tangent along the ray index
    cosang = (tx * np.cos(th) + ty * np.sin(th)) / np.hypot(tx, ty)
    assert np.abs(cosang[2:-2]).max() < 0.02


# ------------------------------------------------------------------ remaining BASELINE configs, per-GPU shard sizes
def test_cfg5_anisotropy_shard_properties(rb, gpu_fields, oracle_fields):
    """cfg5 shard (anisotropy gamma=3, op11, 131 072 rays = 1/8 of 1 048 576): the horizontal momentum p_x is the
    ray parameter of a vertically heterogeneous medium and must be conserved (reference metric, :1354-1360);
    a 1/1024 subsample is compared with the oracle."""
    from oracle import rt_oracle as O
    R = 131072
    th = np.linspace(0, np.pi / 2, R)
    lim = LIMITS["anisotropy"]
    ms = int(np.ceil(80 / rb.DELTA_S) + 1)
    b = rb.Batch(gpu_fields("anisotropy"), 11, rb.DELTA_S, ms, lim, 3, th, -2.0, -2.0, record_stride=0)
    b.run()
    d, fin, st = b.d_ray(), b.final(), b.stats()
    b.close()
    assert st["ray_steps"] == int(d[2].sum()) and 1100 <= d[2].min() and d[2].max() <= 2900
    n0 = 0.07142864686293911
    coef0 = np.sqrt((3 * np.sin(th)) ** 2 + np.cos(th) ** 2)
    px0 = n0 * coef0 * np.cos(th) * (1 + (-np.sin(th) ** 2) * 8 / coef0 ** 2)
    assert np.max(np.abs(fin[6] - px0)) / n0 < 5e-4               # drift relative to p_x's scale (p_x -> 0 at pi/2)
    sub = slice(0, R, 1024)
    o = O.trazar(oracle_fields("anisotropy"), 11, 3, rb.DELTA_S, ms, lim, -2.0, -2.0, th[sub], record_stride=0, nthreads=8)
    assert np.array_equal(d[2][sub], o["d_ray"][2])
    assert np.array_equal(fin[:, sub], o["final"])               # every sampled ray: the oracle's bits


def test_cfg4_fp32_shard_properties(rb, gpu_fields):
    """cfg4 shard (vert_heterogeneous, fp32 state + field, 1 048 576 rays = 1/8 of 8 388 608).  The reference is
    fp64-only; measured against this library's fp64 path on a 1/64 subsample: same step counts within one row,
    end points within 2e-5 (fp64 accumulators under the fp32 arithmetic), p_x conserved to 1e-3."""
    R = 1 << 20
    th = np.linspace(0, np.pi / 2, R)
    lim = LIMITS["vert_heterogeneous"]
    ms = int(np.ceil(80 / rb.DELTA_S) + 1)
    b = rb.Batch(gpu_fields("vert_heterogeneous", 1), 6, rb.DELTA_S, ms, lim, 1, th, -2.0, -2.0, record_stride=0)
    b.run()
    d32, f32 = b.d_ray(), b.final()
    b.close()
    sub = slice(0, R, 64)
    a = rb.Batch(gpu_fields("vert_heterogeneous", 0), 6, rb.DELTA_S, ms, lim, 1, th[sub], -2.0, -2.0, record_stride=0)
    a.run()
    d64, f64 = a.d_ray(), a.final()
    a.close()
    assert np.max(np.abs(d32[2][sub] - d64[2])) <= 1
    same = d32[2][sub] == d64[2]
    err = np.abs(f32[:2, sub] - f64[:2])[:, same].max()
    print(f"cfg4 fp32 vs fp64 end points: {err:.2e} on {same.sum()} rays with equal step count")
    assert same.mean() > 0.99 and err < 2e-5
    n0 = 0.07142864686293911
    assert np.max(np.abs(f32[6] - n0 * np.cos(th))) / n0 < 1e-3


def test_device_tensor_views_alias_library_memory(rb, gpu_fields):
    """Batch.device_tensors(): zero-copy torch views (rtmi_batch_view) used for RCCL read-back."""
    import torch
    th = np.linspace(0, np.pi / 2, 777)
    b = rb.Batch(gpu_fields("vert_heterogeneous"), 6, rb.DELTA_S, 30228, LIMITS["vert_heterogeneous"], 1, th, -2.0, -2.0,
                 record_stride=32)
    b.run()
    t = b.device_tensors()
    fin, d = b.final(), b.d_ray()
    assert t["x"].is_cuda and t["s_ray"].shape == (b.rec_rows, 6, 777)
    assert np.array_equal(t["x"].cpu().numpy(), fin[0]) and np.array_equal(t["T"].cpu().numpy(), fin[8])
    assert np.array_equal(t["istep"].cpu().numpy(), d[2].astype(np.int32))
    s, n = b.rows(want_n_ray=True)
    assert np.array_equal(t["s_ray"].cpu().numpy(), s) and np.array_equal(t["n_ray"].cpu().numpy(), n)
    assert int(t["istep"].sum().item()) == b.stats()["ray_steps"]      # reductions can stay on the device
    del t
    b.close()


def test_bad_launch_conditions_do_not_disturb_neighbours(rb, gpu_fields):
    """NaN / infinite launch angles and positions far outside the grid: those rays produce NaN or clamped-field
    rows and stop at max_size, every other ray of the batch is bit-identical to a clean batch, nothing hangs."""
    F = gpu_fields("vert_heterogeneous")
    lim = LIMITS["vert_heterogeneous"]
    th = np.array([0.3, np.nan, np.inf, 0.5, 0.7, 1e12, 0.9])
    x0 = np.array([-2.0, -2.0, -2.0, -2.0, 1e6, -2.0, np.nan])
    for mode, path in (("plain", 2), ("plain", 1), ("refill", 2)):
        b = rb.Batch(F, 6, rb.DELTA_S, 600, lim, 1, th, x0, -2.0, record_stride=1, launch_mode=mode, field_path=path)
        b.run()
        d, fin, rows = b.d_ray(), b.final(), b.rows()
        b.close()
        c = rb.Batch(F, 6, rb.DELTA_S, 600, lim, 1, th[[0, 3]], -2.0, -2.0, record_stride=1)
        c.run()
        assert np.array_equal(fin[:, [0, 3]], c.final()) and np.array_equal(rows[:, :, [0, 3]], c.rows())
        c.close()
        assert d[2, 1] == 599 and d[2, 2] == 599 and np.isnan(fin[0, 1])      # never "outside": NaN compares false
        assert d[2, 4] == 1                                                    # launched far outside: one step, stored (Q7)
    from raytracing_amd._lib import RtmiError
    with pytest.raises(RtmiError, match="R must be"):
        rb.Batch(F, 6, rb.DELTA_S, 600, lim, 1, np.array([]), -2.0, -2.0)      # empty batch: rejected, not launched


@pytest.mark.parametrize("scen,m,mode", [("vert_heterogeneous", 6, "plain"), ("interface", 7, "plain"), ("vert_heterogeneous", 6, "refill")])
def test_sort_rays_answers_in_caller_order(scen, m, mode, rb, gpu_fields, oracle_fields):
    """sort_rays=1 reorders rays inside the batch (coherent lanes) but every read -- d_ray, final state, rows,
    n_ray, metrics, isochrones, set_state -- answers in the caller's order, bit-identical to the unsorted batch."""
    rng = np.random.default_rng(11)
    R = 1500
    lim = LIMITS[scen]
    th = rng.permutation(np.linspace(0.06, np.pi / 2, R))
    x0 = np.where(rng.random(R) < 0.2, -1.0, -2.0)          # two launch points -> two cell blocks in the sort key
    th[7] = np.nan
    ms = 3000
    out = []
    for srt in (False, True):
        b = rb.Batch(gpu_fields(scen), m, rb.DELTA_S, ms, lim, 1, th, x0, -2.0, record_stride=1, sort_rays=srt,
                     launch_mode=mode)
        b.step(50)
        b.run()
        s, n = b.rows(1000, 700, want_n_ray=True)
        perm = b.device_tensors().get("perm")
        out.append((b.d_ray(), b.final(), s, n, b.metric("px_cv"), b.metric("closure"), b.isochrones([0.05, 0.1]),
                    None if perm is None else perm.cpu().tolist()))      # copied before the batch (and its memory) goes
        del perm
        b.close()
    assert out[0][7] is None and sorted(out[1][7]) == list(range(R))
    for u, v in zip(out[0][:7], out[1][:7]):
        assert np.array_equal(u, v, equal_nan=True)
    # the sorted batch against the oracle directly, in the caller's ray order (ray 7 has a NaN launch angle)
    from oracle import rt_oracle as O
    o = O.trazar(oracle_fields(scen), m, 1, rb.DELTA_S, ms, lim, x0, -2.0, th, record_stride=1, nthreads=8)
    ok = np.ones(R, bool); ok[7] = False
    d, fin, s = out[1][0], out[1][1], out[1][2]
    assert np.array_equal(d[2][ok], o["d_ray"][2][ok])
    tol = REL
    assert relerr(fin[:, ok], o["final"][:, ok]) < tol
    assert relerr(s[:, :, ok], o["s_ray"][1000:1700][:, :, ok]) < tol


def test_north_star_1m_rays_vs_oracle_subsample(rb, gpu_fields, oracle_fields):
    """The bench workload itself (vert_heterogeneous, 1 048 576 rays, op6, default DELTA_S, fp64): every 512th ray
    against the oracle (end state to 1e-9, step counts exactly), the device step counter against sum(d_ray[2]),
    and p_x conservation over the whole fan."""
    from oracle import rt_oracle as O
    R = 1 << 20
    th = np.linspace(0, np.pi / 2, R)
    lim = LIMITS["vert_heterogeneous"]
    ms = int(np.ceil(80 / rb.DELTA_S) + 1)
    b = rb.Batch(gpu_fields("vert_heterogeneous"), 6, rb.DELTA_S, ms, lim, 1, th, -2.0, -2.0, record_stride=0)
    b.run()
    d, fin, st = b.d_ray(), b.final(), b.stats()
    b.close()
    assert st["ray_steps"] == int(d[2].sum()) == 1937541698           # the figure bench.py divides by the time
    sub = slice(0, R, 512)
    o = O.trazar(oracle_fields("vert_heterogeneous"), 6, 1, rb.DELTA_S, ms, lim, -2.0, -2.0, th[sub], record_stride=0,
                 nthreads=8)
    assert np.array_equal(d[2][sub], o["d_ray"][2])
    err = relerr(fin[:, sub], o["final"])
    print(f"1M-ray fan, 2048-ray subsample vs oracle: max rel err {err:.2e}")
    assert err < REL and relerr(d[:2, sub], o["d_ray"][:2]) < REL
    n0 = 0.07142864686293911
    assert np.max(np.abs(fin[6] - n0 * np.cos(th))) / n0 < 5e-4          # the scheme's own p_x drift (CV threshold scale)


def test_headline_bench_configuration_every_row_and_whole_record_checksum(rb, gpu_fields, oracle_fields):
    """The configuration bench.py times, exactly (vert_heterogeneous, 1 048 576 rays, op6, fp64, full record in
    s_ray[3072][6][R] = 151 GB, no n_ray, lazy_clear, the library's default schedule RTMI_LAUNCH_AUTO, 512-step slices):
      * every recorded row of every 512th ray against the oracle's s_ray at 1e-9 (what the rows must hold: RT_bench.py:871-875),
        rows past a ray's last one zero (np.zeros, :802);
      * RTMI_LAUNCH_AUTO really runs the time-sliced kernel first and the plain launch on the re-run, and the two leave the
        same 151 GB: a per-quantity checksum of the whole record (the bits summed as 64-bit integers, wrap-around, so the
        order of summation does not matter) is equal between them, after the record was zeroed in between."""
    import torch
    from oracle import rt_oracle as O
    R, rows = 1 << 20, 3072
    th = np.linspace(0, np.pi / 2, R)
    lim = LIMITS["vert_heterogeneous"]
    ms = int(np.ceil(80 / rb.DELTA_S) + 1)
    b = rb.Batch(gpu_fields("vert_heterogeneous"), 6, rb.DELTA_S, ms, lim, 1, th, -2.0, -2.0, record_stride=1, rec_rows=rows,
                 lazy_clear=True, keep_n_ray=False)          # launch_mode, slice_steps: rtmi_params' defaults
    b.run()
    st = b.stats()
    assert st["launch_mode_used"] == "sliced" and st["ray_steps"] == 1937541698 and st["live_rays"] == 0
    assert st["auto_fallbacks"] == 0                          # no time-sliced launch gave up a wait (rtmi_stats, ABI v6)
    s = b.device_tensors()["s_ray"]
    assert tuple(s.shape) == (rows, 6, R) and s.dtype == torch.float64

    def checksum():
        return [int(v) for v in s.view(torch.int64).sum(dim=(0, 2)).cpu()]

    sub = slice(0, R, 512)
    got = s[:, :, sub].cpu().numpy()                          # 3072 x 6 x 2048 values: 302 MB of the 151 GB
    d = b.d_ray()
    o = O.trazar(oracle_fields("vert_heterogeneous"), 6, 1, rb.DELTA_S, ms, lim, -2.0, -2.0, th[sub], record_stride=1,
                 rec_rows=rows, nthreads=8)
    assert np.array_equal(d[2][sub], o["d_ray"][2])
    err = relerr(got, o["s_ray"])
    print(f"headline configuration: {got.shape[0]} rows x 6 x {got.shape[2]} rays vs oracle: max rel err {err:.2e}")
    assert err < REL
    last = d[2][sub].astype(int)
    for k in range(got.shape[2]):
        assert not got[last[k] + 1:, :, k].any()             # rows beyond the last written one stay zero
    c_sliced = checksum()
    s.zero_()                                                 # lazy_clear: reset itself does not clear; make the re-run prove itself
    torch.cuda.synchronize()
    b.reset(); b.run()
    st2 = b.stats()
    assert st2["launch_mode_used"] == "plain" and st2["ray_steps"] == st["ray_steps"]
    c_plain = checksum()
    assert c_sliced == c_plain, (c_sliced, c_plain)
    assert np.array_equal(s[:, :, sub].cpu().numpy(), got)
    for _ in range(3):                                        # two more exploration runs, then the faster: same record again
        b.reset(); b.run()
    assert b.stats()["launch_mode_used"] in ("sliced", "plain") and checksum() == c_sliced and b.stats()["auto_fallbacks"] == 0
    b.close()


@pytest.mark.parametrize("scen,m", [("vert_heterogeneous", 6), ("fisheye", 2), ("interface", 8), ("vert_heterogeneous", 3),
                                    ("fisheye", 7), ("vert_heterogeneous", 1), ("interface", 4), ("fisheye", 9),
                                    ("anisotropy", 11), ("vert_heterogeneous", 5), ("anisotropy", 10)])
def test_random_rays_through_grid_ends_vs_oracle(scen, m, rb, gpu_fields, oracle_fields):
    """Seeded random launch points and directions anywhere on the padded grid, with the box opened up to the grid's
    rim: rays cross the not-a-knot end cells (general fpbspl path, clamped lookups) and hit every fix-up branch of
    the cell search.  Compared with the oracle ray by ray."""
    from oracle import rt_oracle as O
    rng = np.random.default_rng(100 + m)
    x, y, *_ = oracle_fields(scen).arrays()
    R = 600 if m not in (5, 9, 10, 11) else 192
    gam = 3 if scen == "anisotropy" else 1
    x0 = rng.uniform(x[0] + 0.05, x[-1] - 0.05, R)
    y0 = rng.uniform(y[0] + 0.05, y[-1] - 0.05, R)
    # a third of the rays start exactly on grid lines / nodes (interval search ties)
    x0[::3] = x[rng.integers(1, len(x) - 1, len(x0[::3]))]
    y0[1::3] = y[rng.integers(1, len(y) - 1, len(y0[1::3]))]
    th = rng.uniform(-np.pi, np.pi, R)
    lim = (x[0] + 0.01, x[-1] - 0.01, y[0] + 0.01, y[-1] - 0.01)
    step, ms = 0.011, 700
    b = rb.Batch(gpu_fields(scen), m, step, ms, lim, gam, th, x0, y0, record_stride=0)
    b.run()
    d, fin = b.d_ray(), b.final()
    b.close()
    o = O.trazar(oracle_fields(scen), m, gam, step, ms, lim, x0, y0, th, record_stride=0, nthreads=8)
    same = d[2] == o["d_ray"][2]
    err = relerr(fin[:, same], o["final"][:, same])
    print(f"{scen} op{m}: {same.sum()}/{R} same step count, max rel err {err:.2e}")
    if m in (3, 4, 5, 7, 9, 10, 11):
        # reference-order methods (op7 by default too) on a field whose coefficients are the oracle's bits: the step count is
        # an integer RESULT and must be the oracle's on every ray
        assert same.all(), f"step counts differ on rays {np.flatnonzero(~same)[:8]}"
        if m != 7:
            assert np.array_equal(fin, o["final"])   # every ray bit-identical
            return
    else:
        # fused forms (~1e-13 from the reference): a ray whose last point lies within that of the box's rim may leave one row
        # apart.  Reported, not dropped: each such ray must differ by exactly one row, and the point where the shorter run
        # stopped must be on the rim to the fused forms' distance from the reference
        for k in np.flatnonzero(~same):
            dk, ok_ = int(d[2, k]), int(o["d_ray"][2, k])
            short = fin[:, k] if dk < ok_ else o["final"][:, k]
            rim = min(abs(short[0] - lim[0]), abs(short[0] - lim[1]), abs(short[1] - lim[2]), abs(short[1] - lim[3]))
            print(f"  ray {k}: {dk} rows here, {ok_} in the oracle; the shorter run stopped {rim:.2e} from the rim")
            assert abs(dk - ok_) == 1 and rim < 1e-9
        assert same.mean() > 0.99
    assert err < REL


@pytest.mark.parametrize("qx,qy", [(8, 8), (12, 10), (40, 17)])
def test_small_custom_grids_vs_oracle(qx, qy, rb):
    """interpolacion() on caller-sampled grids so small that every cell touches the not-a-knot ends (the regime rounds 1-2's
    LDS tile and uniform-knot fast path excluded; the cell polynomials have no such case).  Coefficients, point lookups
    (inside and outside the grid) and short traces against the oracle."""
    from oracle import rt_oracle as O
    rng = np.random.default_rng(qx * 100 + qy)
    x = np.linspace(-1.0, 2.0, qx); y = np.linspace(0.5, 2.5, qy)
    X, Y = np.meshgrid(x, y)
    Z = 1.2 + 0.3 * np.sin(1.3 * X) * np.cos(0.7 * Y) + 0.05 * rng.standard_normal(X.shape)
    delta = 0.11
    F = rb.Field.from_samples(x, y, Z, delta)
    OF = O.Field.from_samples(x, y, Z, delta)
    for a, b in zip(F.arrays()[2:], OF.arrays()[2:]):
        assert np.abs(a - b).max() <= 1e-12 * np.abs(b).max()
    px = rng.uniform(-1.5, 2.5, 400); py = rng.uniform(0.0, 3.0, 400)
    for u, v in zip(F.n_gradient(px, py), OF.n_gradient(px, py)):
        assert np.abs(u - v).max() <= 1e-12 * max(np.abs(v).max(), 1.0)
    R = 130
    x0 = rng.uniform(-0.9, 1.9, R); y0 = rng.uniform(0.6, 2.4, R); th = rng.uniform(-np.pi, np.pi, R)
    lim = (-0.99, 1.99, 0.51, 2.49)
    for m in (2, 6):
        for path in (1, 2):
            b = rb.Batch(F, m, 0.01, 500, lim, 1, th, x0, y0, record_stride=0, field_path=path)
            b.run()
            d, fin = b.d_ray(), b.final()
            b.close()
            o = O.trazar(OF, m, 1, 0.01, 500, lim, x0, y0, th, record_stride=0)
            same = d[2] == o["d_ray"][2]
            assert same.mean() > 0.98
            assert relerr(fin[:, same], o["final"][:, same]) < 1e-8      # noisy Z: gradients are O(1) per cell
    F.close()


def test_per_ray_step_batch_equals_separate_batches(rb, gpu_fields):
    """rtmi_batch_set_per_ray: rays of one batch with different DELTA_S / max_size are bit-identical to the same
    rays traced in separate uniform batches (rows, d_ray, final state), also after reset and with sort_rays."""
    F = gpu_fields("vert_heterogeneous")
    lim = LIMITS["vert_heterogeneous"]
    th = np.linspace(0.1, 1.4, 40)
    cands = [(rb.DELTA_S, 700), (0.01, 300), (0.05, 60)]
    ref = []
    for st, ms in cands:
        b = rb.Batch(F, 7, st, ms, lim, 1, th, -2.0, -2.0, record_stride=1, rec_rows=700)
        b.run()
        ref.append((b.d_ray(), b.final(), b.rows()))
        b.close()
    for srt in (False, True):
        b = rb.Batch(F, 7, cands[0][0], 700, lim, 1, np.tile(th, 3), -2.0, -2.0, record_stride=1, sort_rays=srt)
        b.set_per_ray(np.repeat([c[0] for c in cands], 40), np.repeat([c[1] for c in cands], 40))
        for attempt in range(2):
            b.run()
            d, fin, rows = b.d_ray(), b.final(), b.rows()
            for i in range(3):
                sl = slice(40 * i, 40 * (i + 1))
                assert np.array_equal(d[:, sl], ref[i][0]) and np.array_equal(fin[:, sl], ref[i][1])
                assert np.array_equal(rows[:, :, sl], ref[i][2])
            b.reset()
        b.close()
    from raytracing_amd._lib import RtmiError
    b = rb.Batch(F, 6, rb.DELTA_S, 100, lim, 1, th, -2.0, -2.0, record_stride=0)
    with pytest.raises(RtmiError, match="max_size"):
        b.set_per_ray(0.01, 101)
    b.close()


CALIBRATED = {"1": {1: 38.64, 2: 38.37, 3: 2.34, 4: 2.53, 5: 2.53, 6: 2.55, 7: 30.05, 8: 2.74, 9: 2.74},      # :1413-1430
              "2": {1: 149, 2: 169, 3: 182, 4: 179, 5: 179, 6: 182, 7: 191, 8: 179, 9: 179}}                   # :1444 (5 % set)


@pytest.mark.parametrize("choice,m", [(c, m) for c in ("1", "2") for m in range(1, 10)])
def test_full_calibration_reproduces_reference_table(choice, m, rb, gpu_fields):
    """The whole DELTA_S search (RT_bench.py:1296-1385) as one candidate x ray launch, for every isotropic method:
    the divisor the find_index rules pick must be the calibrated value the reference hard-codes -- the interface
    table at :1413-1430 and the fisheye's 5 %-closure set quoted at :1444.  For interface op1/op2/op7 the reference
    says the search interval has to be narrowed around the recorded value (:1415-1420); +-1 is used here."""
    scen = {"1": "interface", "2": "fisheye"}[choice]
    want = CALIBRATED[choice][m]
    div, opt = rb.delta_s_candidates(choice)
    if choice == "1" and m in (1, 2, 7):
        div = np.arange(want + 1.0, want - 1.0 - rb.DELTA_STEP, -rb.DELTA_STEP)
        opt = rb.SIGMA / div
    res = rb.search_delta_sweep(rb.METHODS[m], gpu_fields(scen), None, opt, div, choice)
    assert len(res) == len(div)
    assert rb.find_divisor(res, div, choice) == want


def test_benchmark_statistic_runs(rb, gpu_fields):
    """rt_bench.benchmark: the reference's protocol (:1516-1541 -- rounds of runs, IQR filter, median of the last
    30 %, stop when two successive medians agree to 0.5 %) over device propagation times of the preset batch."""
    F = gpu_fields("vert_heterogeneous")
    z, grd = rb.FieldSpline(F, "n"), (rb.FieldSpline(F, "dy"), rb.FieldSpline(F, "dx"))
    t = rb.benchmark(rb.op6, z, grd, rb.DELTA_S, 91, "3", trial=8, replicas=2, max_rounds=6)
    assert 1e-5 < t < 0.5          # 31 rays, ~2 900 sequential steps: a few milliseconds of device time


def test_isochrones_on_truncated_record(rb, gpu_fields):
    """rec_rows < max_size is allowed (rows beyond are simply not kept): the isochrone stage must then work on the
    rows that exist and never read past the allocation.  A ray that outruns rec_rows answers from its
    first rec_rows rows: times it reaches within them match the full record, later times are NaN."""
    F = gpu_fields("vert_heterogeneous")
    lim = LIMITS["vert_heterogeneous"]
    th = np.linspace(0.2, 1.4, 64)
    full = rb.Batch(F, 6, rb.DELTA_S, 30228, lim, 1, th, -2.0, -2.0, record_stride=1, rec_rows=3072)
    cut = rb.Batch(F, 6, rb.DELTA_S, 30228, lim, 1, th, -2.0, -2.0, record_stride=1, rec_rows=900)
    full.run(); cut.run()
    assert full.d_ray()[2].min() > 900                       # every ray outruns the truncated record
    times = np.array([0.02, 0.05, 0.08, 0.3])
    a, b = full.isochrones(times), cut.isochrones(times)
    T_end = cut.rows(899, 1)[0, 4]                           # traveltime of the last kept row, per ray
    for it, t in enumerate(times):
        inside = T_end >= t
        assert np.array_equal(np.isnan(b[it, 0]), ~inside)
        # away from the cut the two agree exactly; at the cut the end-point derivative rule applies to different rows
        far = inside & (T_end - t > 5 * 0.07 * rb.DELTA_S)
        assert np.array_equal(b[it][:, far], a[it][:, far])
        if inside.any():
            assert np.abs(b[it][:, inside] - a[it][:, inside]).max() < 1e-6
    assert np.isnan(b[3]).all() and not np.isnan(a[3]).all()
    full.close(); cut.close()


def test_set_per_ray_only_on_a_fresh_batch_and_reset_clears_rows(rb, gpu_fields):
    """rtmi_batch_set_per_ray after stepping would revive finished rays: RTMI_ERR_STATE.
    rtmi_batch_reset restores the reference's np.zeros invariant (rows past the last written row read 0) at once;
    lazy_clear=1 defers it to the re-run, which rewrites the same rows."""
    from raytracing_amd._lib import RtmiError
    F = gpu_fields("vert_heterogeneous")
    lim = LIMITS["vert_heterogeneous"]
    th = np.linspace(0.1, 1.4, 40)
    b = rb.Batch(F, 6, rb.DELTA_S, 400, lim, 1, th, -2.0, -2.0, record_stride=1)
    b.set_per_ray(0.01, 300)                                  # fresh: fine
    b.step(5)
    with pytest.raises(RtmiError, match="fresh or reset"):
        b.set_per_ray(0.02, 200)
    b.run()
    first = b.rows()
    assert first[1:].any()
    b.reset()
    assert not b.rows()[1:].any() and np.array_equal(b.rows()[0], first[0])     # zeros again, row 0 rewritten
    b.set_per_ray(0.01, 300)                                  # reset: fine again
    b.run()
    assert np.array_equal(b.rows(), first)
    b.close()
    lz = rb.Batch(F, 6, 0.01, 300, lim, 1, th, -2.0, -2.0, record_stride=1, lazy_clear=True)
    lz.run()
    ref = lz.rows()
    lz.reset()
    assert np.array_equal(lz.rows()[5:], ref[5:])            # lazy: the previous pass's rows are still there ...
    lz.run()
    assert np.array_equal(lz.rows(), ref)                     # ... and the re-run reproduces them exactly
    lz.close()


# ------------------------------------------------------------------ the ill-conditioned rays of the interface scenario
@pytest.mark.parametrize("method, first", [(6, 487296), (8, 487168), (1, 483328), (2, 483328)])
def test_critical_rays_of_the_1m_interface_fan(method, first, rb, gpu_fields, oracle_fields):
    """Where the 1 048 576-ray interface fan splits into reflected and refracted rays (near 45 degrees; each method at its own
    angle) a ray runs along the interface and amplifies last-bit differences a million times: in the fused forms ALONE a handful
    of rays per method -- 2 to 6 of the 1 M (tools/critical_ray_window.py) -- leave 1e-9, p and theta on rows inside the interface.
    Those rays are ill-conditioned in the REFERENCE (its own rows move 300 times further when the launch angle changes by 1e-12
    of itself), so only its own roundings reproduce it there -- and a DEFAULT batch now finds such rays on the way (the hover
    sum, rt_device.h) and re-traces them in reference order by itself (rtmi_params.no_retrace).  1 024 contiguous rays of the fan
    around the split, every 16th row: (a) rtmi_params.reference_order = 1 gives the oracle's bits on every one of them; (b) a
    default batch is within 1e-9 of the oracle on EVERY ray -- none excepted -- with equal step counts, and has re-traced some;
    (c) with no_retrace = 1 (round 4's default) the known offenders are back: the re-trace is what closes them."""
    from oracle import rt_oracle as O
    R, W = 1 << 20, 1024
    th = np.linspace(2 * np.pi / 60, np.pi / 2, R)[first:first + W]
    lim = LIMITS["interface"]
    ms = int(np.ceil(80 / rb.DELTA_S) + 1)
    kw = dict(record_stride=16, rec_rows=600)

    def groups(a, w):
        return np.array([np.abs(a[:, q] - w[:, q]).max(axis=(0, 1)) / np.abs(w[:, q]).max() for q in ([0, 1], [2, 3], [4], [5])]).max(axis=0)

    OF = oracle_fields("interface")
    o = O.trazar(OF, method, 1, rb.DELTA_S, ms, lim, -2.0, -2.0, th, nthreads=16, **kw)
    assert np.ptp(o["final"][1]) > 1.0                                   # the window does hold the split: rays leave at different heights
    b = rb.Batch(gpu_fields("interface"), method, rb.DELTA_S, ms, lim, 1, th, -2.0, -2.0, reference_order=True, **kw)
    b.run()
    assert np.array_equal(b.rows(), o["s_ray"]) and np.array_equal(b.final(), o["final"]) and np.array_equal(b.d_ray(), o["d_ray"])
    b.close()
    b = rb.Batch(gpu_fields("interface"), method, rb.DELTA_S, ms, lim, 1, th, -2.0, -2.0, **kw)
    b.run()
    s, d, fin, st = b.rows(), b.d_ray(), b.final(), b.stats()
    b.close()
    assert np.array_equal(d[2], o["d_ray"][2])
    dev = groups(s, o["s_ray"])
    print(f"op{method} default: {st['retraced']} of {W} rays re-traced in reference order; largest difference from the oracle {dev.max():.1e} (rows), "
          f"{relerr(fin, o['final']):.1e} (final state)")
    assert (dev > REL).sum() == 0 and dev.max() < 1e-10                    # every ray, no waiver (measured: 1.2e-11)
    assert relerr(fin, o["final"]) < REL and relerr(d[:2], o["d_ray"][:2]) < REL
    assert 0 < st["retraced"] < W and st["retrace_overflow"] == 0
    b = rb.Batch(gpu_fields("interface"), method, rb.DELTA_S, ms, lim, 1, th, -2.0, -2.0, retrace=False, **kw)
    b.run()
    s0, st0 = b.rows(), b.stats()
    b.close()
    dev0 = groups(s0, o["s_ray"])
    print(f"op{method} no_retrace: {int((dev0 > REL).sum())} rays beyond 1e-9 (largest {dev0.max():.1e})")
    assert st0["retraced"] == 0 and (dev0 > REL).sum() >= 1               # the window was chosen around the known offenders


@pytest.mark.parametrize("knob", ["RTMI_RETRACE_CUS", "RTMI_NO_DISPATCH_ORDER"])
def test_retrace_knobs_change_no_bit(knob, rb, gpu_fields, monkeypatch):
    """Where the re-trace runs (compute units of its own behind CU masks, the main kernel on a stream of the batch's own) and in
    which order a re-run batch's bundles are dispatched are scheduling: with RTMI_RETRACE_CUS=0 (no compute units set aside, the
    main kernel on the caller's stream) and with RTMI_NO_DISPATCH_ORDER=1 every row, final state and d_ray are the default's bits,
    on a batch's first run and on a re-run, and the same rays are re-traced."""
    R = 1 << 18                                  # 1 024 bundles: the plain kernel (k_advance), whose block order the batch rotates
    th = np.linspace(2 * np.pi / 60, np.pi / 2, R)
    lim = LIMITS["interface"]
    ms = int(np.ceil(80 / rb.DELTA_S) + 1)
    kw = dict(record_stride=256, rec_rows=40, launch_mode="plain")

    def two_runs():
        b = rb.Batch(gpu_fields("interface"), 6, rb.DELTA_S, ms, lim, 1, th, -2.0, -2.0, **kw)
        out = []
        for _ in range(2):
            b.run()
            out.append((b.rows(), b.final(), b.d_ray(), b.stats()))
            b.reset()
        b.close()
        return out

    ref = two_runs()
    assert ref[0][3]["retraced"] > 0 and ref[1][3]["dispatch_first"] > 0
    monkeypatch.setenv(knob, "0" if knob == "RTMI_RETRACE_CUS" else "1")
    alt = two_runs()
    if knob == "RTMI_NO_DISPATCH_ORDER":
        assert alt[1][3]["dispatch_first"] == 0
    for a, c in zip(ref, alt):
        assert a[3]["retraced"] == c[3]["retraced"]
        for x, y in zip(a[:3], c[:3]):
            assert np.array_equal(x, y)
    for x, y in zip(ref[0][:3], ref[1][:3]):      # ... and the re-run, rotated, gives the first run's bits
        assert np.array_equal(x, y)


def test_a_rerun_batch_dispatches_its_critical_bundles_first(rb, gpu_fields):
    """A batch that handed critical rays over remembers which 256-ray bundles held them and its next runs start there
    (rtmi_stats.dispatch_first; the plain kernel's block order is rotated, nothing else): the same rays re-traced, every row,
    final state and d_ray the same bits as the first run's."""
    R = 1 << 16
    th = np.linspace(2 * np.pi / 60, np.pi / 2, R)
    lim = LIMITS["interface"]
    ms = int(np.ceil(80 / rb.DELTA_S) + 1)
    b = rb.Batch(gpu_fields("interface"), 6, rb.DELTA_S, ms, lim, 1, th, -2.0, -2.0, record_stride=64, rec_rows=150, launch_mode="plain")
    b.run()
    first = (b.rows(), b.final(), b.d_ray(), b.stats())
    assert first[3]["retraced"] > 0 and first[3]["dispatch_first"] > 0
    # the split of this fan is near 45 degrees: ray 0.464 R, bundle 0.464 R / 256
    assert abs(first[3]["dispatch_first"] - 0.464 * R / 256) < 4
    for _ in range(2):
        b.reset()
        b.run()
        again = (b.rows(), b.final(), b.d_ray(), b.stats())
        assert again[3]["retraced"] == first[3]["retraced"] and again[3]["dispatch_first"] == first[3]["dispatch_first"]
        for a, c in zip(first[:3], again[:3]):
            assert np.array_equal(a, c)
    b.close()


@pytest.mark.parametrize("tilt_deg, method", [(3.0, 6), (11.0, 6), (11.0, 1)])
def test_critical_rays_of_a_tilted_wall(tilt_deg, method, rb):
    """The hand-over must not lean on the interface scenario's wall lying along a grid line: the same sigmoid wall tilted against
    the grid (both sides get the SAMPLES: rtmi_field_from_samples / the oracle's from_samples), where the spline's gradient
    direction wobbles from cell to cell.  The split of the 1 048 576-ray fan is found on the device (every 64th ray), then the
    1 024 contiguous rays around it are compared with the oracle, every 16th row: a default batch within 1e-9 on EVERY ray, equal
    step counts, some rays re-traced.  (Round 5's first hover criterion -- an angle threshold of 0.02 rad -- left 3-5 rays of
    4 096 beyond 1e-9 at 11 degrees: tools/tilted_interface_probe.py, profiles/r05_tilted_walls.txt.)"""
    from oracle import rt_oracle as O
    R, W = 1 << 20, 1024
    lim = LIMITS["interface"]
    ms = int(np.ceil(80 / rb.DELTA_S) + 1)
    x, y = O.Field("interface", lim, rb.DELTA).arrays()[:2]
    X, Y = np.meshgrid(x, y)
    a = np.radians(tilt_deg)
    d = -np.sin(a) * (X + 2.0) + np.cos(a) * Y
    Z = np.sqrt(2.0) - (np.sqrt(2.0) - 1.0) / (1.0 + np.exp(-np.clip(d / 0.005, -700, 700)))
    F = rb.Field.from_samples(x, y, Z, rb.DELTA)
    OF = O.Field.from_samples(x, y, Z, rb.DELTA)
    fan = np.linspace(2 * np.pi / 60, np.pi / 2, R)
    b = rb.Batch(F, method, rb.DELTA_S, ms, lim, 1, fan[::64], -2.0, -2.0, record_stride=0)
    b.run()
    fin = b.final()
    b.close()
    k = int(np.argmax(np.abs(np.diff(fin[1])) + np.abs(np.diff(fin[0]))))
    i0 = min(max(0, k * 64 + 32 - W // 2), R - W)
    th = fan[i0:i0 + W]
    kw = dict(record_stride=16, rec_rows=600)
    o = O.trazar(OF, method, 1, rb.DELTA_S, ms, lim, -2.0, -2.0, th, nthreads=16, **kw)
    assert np.ptp(o["final"][1]) > 1.0
    out = {}
    for retrace in (True, False):
        b = rb.Batch(F, method, rb.DELTA_S, ms, lim, 1, th, -2.0, -2.0, retrace=retrace, **kw)
        b.run()
        s, dr, st = b.rows(), b.d_ray(), b.stats()
        b.close()
        same = dr[2] == o["d_ray"][2]
        dev = np.array([np.abs(s[:, q][:, :, same] - o["s_ray"][:, q][:, :, same]).max(axis=(0, 1)) / np.abs(o["s_ray"][:, q]).max()
                        for q in ([0, 1], [2, 3], [4], [5])]).max(axis=0)
        out[retrace] = (int(same.sum()), dev, st)
    F.close()
    print(f"wall tilted {tilt_deg:g} deg, op{method}, split at ray {k * 64 + 32}: default {out[True][2]['retraced']} rays re-traced, largest "
          f"{out[True][1].max():.1e}; no_retrace: {int((out[False][1] > REL).sum())} rays beyond 1e-9 (largest {out[False][1].max():.1e})")
    n, dev, st = out[True]
    assert n == W and (dev > REL).sum() == 0 and dev.max() < 2e-10
    assert 0 < st["retraced"] < W and st["retrace_overflow"] == 0


@pytest.mark.parametrize("method", [6, 1, 2, 8])
def test_interface_rays_do_not_depend_on_their_wave_mates(method, rb, gpu_fields):
    """A ray's bits must not depend on which rays share its wave: the wave-level layouts of the fused step -- the flat-cell map's
    early answer when every live lane sits in one flat cell, the wave-uniform cell through the scalar cache, the votes of the
    small-angle forms -- only pick code paths, never values.  The sorted interface fan (coherent waves) against the same rays
    shuffled (every wave a mix of rays in different cells, flat and not): every recorded row, the final state and d_ray equal,
    ray by ray."""
    R = 4096
    th = np.linspace(2 * np.pi / 60, np.pi / 2, R)
    perm = np.random.default_rng(11).permutation(R)
    lim = LIMITS["interface"]
    ms = int(np.ceil(80 / rb.DELTA_S) + 1)
    kw = dict(record_stride=64, rec_rows=160, launch_mode="plain", retrace=False)
    a = rb.Batch(gpu_fields("interface"), method, rb.DELTA_S, ms, lim, 1, th, -2.0, -2.0, **kw)
    a.run()
    ra, fa, da = a.rows(), a.final(), a.d_ray()
    a.close()
    b = rb.Batch(gpu_fields("interface"), method, rb.DELTA_S, ms, lim, 1, th[perm], -2.0, -2.0, **kw)
    b.run()
    rb_, fb, db = b.rows(), b.final(), b.d_ray()
    b.close()
    assert np.array_equal(db, da[:, perm]) and np.array_equal(fb, fa[:, perm]) and np.array_equal(rb_, ra[:, :, perm])


@pytest.mark.parametrize("mode", ["plain", "sliced", "refill", "steps"])
def test_retrace_is_the_same_in_every_schedule(mode, rb, gpu_fields):
    """The automatic re-trace of critical rays is a property of the RAY (its own hover sum): the same rays are handed over and
    the same bits come back whichever schedule runs the fused part -- the plain launch, time-sliced bundles, lane refill, or a host
    that steps the batch 300 rows at a time (rays handed over are re-traced at the first read; rtmi_params.no_retrace).  4 096
    contiguous rays of the 1 M-ray interface fan around op6's split against the plain launch."""
    R = 1 << 20
    th = np.linspace(2 * np.pi / 60, np.pi / 2, R)[485704:485704 + 4096]
    lim = LIMITS["interface"]
    ms = int(np.ceil(80 / rb.DELTA_S) + 1)
    kw = dict(record_stride=16, rec_rows=600)

    def run(launch_mode, stepped=False):
        b = rb.Batch(gpu_fields("interface"), 6, rb.DELTA_S, ms, lim, 1, th, -2.0, -2.0, launch_mode=launch_mode, **kw)
        if stepped:
            while True:
                b.step(300)
                if b.stats()["live_rays"] == 0:
                    break
        else:
            b.run()
        out = (b.rows(), b.final(), b.d_ray(), b.stats())
        b.close()
        return out
    ref = run("plain")
    assert ref[3]["retraced"] > 100 and ref[3]["retrace_overflow"] == 0
    if mode == "plain":
        return
    got = run("plain" if mode == "steps" else mode, stepped=mode == "steps")
    assert got[3]["retraced"] == ref[3]["retraced"]
    assert np.array_equal(got[2], ref[2]) and np.array_equal(got[1], ref[1]) and np.array_equal(got[0], ref[0])
    assert got[3]["ray_steps"] == ref[3]["ray_steps"] == int(ref[2][2].sum())


def test_retrace_survives_checkpoint_and_resume(rb, gpu_fields):
    """get_state in the middle of the hover (some rays already handed over and finished, others with half a hover sum), restore
    into a fresh batch, run on: the same bits as one uninterrupted run (the hover sum travels in aux4 row 2)."""
    R = 1 << 20
    th = np.linspace(2 * np.pi / 60, np.pi / 2, R)[487752 - 512:487752 + 512]
    lim = LIMITS["interface"]
    ms = int(np.ceil(80 / rb.DELTA_S) + 1)
    F = gpu_fields("interface")
    a = rb.Batch(F, 6, rb.DELTA_S, ms, lim, 1, th, -2.0, -2.0, record_stride=0)
    a.run()
    fa, da, sa = a.final(), a.d_ray(), a.stats()
    a.close()
    assert sa["retraced"] > 50
    b = rb.Batch(F, 6, rb.DELTA_S, ms, lim, 1, th, -2.0, -2.0, record_stride=0)
    b.step(1250)                                      # the rays are inside the transition band, hovering
    st9, aux4, istep, alive = b.get_state()
    assert aux4[2].max() > 0                          # hover sums under way
    b.close()
    c = rb.Batch(F, 6, rb.DELTA_S, ms, lim, 1, th, -2.0, -2.0, record_stride=0)
    c.restore_state(st9, aux4, istep, alive)
    c.run()
    fc, dc = c.final(), c.d_ray()
    c.close()
    assert np.array_equal(dc, da) and np.array_equal(fc, fa)


# ------------------------------------------------------------------ BASELINE configs at FULL size on one GPU
def test_cfg5_anisotropy_full_1m_rays(rb, gpu_fields, oracle_fields):
    """cfg5 whole: anisotropy (gamma = 3), op11, 1 048 576 rays fp64 on ONE MI355X, every 16th row recorded (9.7 GB).  p_x (the
    ray parameter of a vertically heterogeneous medium) conserved over the whole fan, the device step counter against
    sum(d_ray[2]), and every 256th ray -- 4 096 rays: d_ray, final state and every recorded row -- against the oracle: the
    oracle's bits (golden-section method, reference-order arithmetic; rows past a ray's last one zero)."""
    from oracle import rt_oracle as O
    R = 1 << 20
    th = np.linspace(0, np.pi / 2, R)
    lim = LIMITS["anisotropy"]
    ms = int(np.ceil(80 / rb.DELTA_S) + 1)
    rows = 3072 // 16
    b = rb.Batch(gpu_fields("anisotropy"), 11, rb.DELTA_S, ms, lim, 3, th, -2.0, -2.0, record_stride=16, rec_rows=rows, keep_n_ray=False)
    b.run()
    d, fin, st = b.d_ray(), b.final(), b.stats()
    sub = slice(0, R, 256)
    got = b.device_tensors()["s_ray"][:, :, sub].cpu().numpy()        # 192 x 6 x 4096 of the record in HBM
    b.close()
    assert st["ray_steps"] == int(d[2].sum()) and st["live_rays"] == 0
    assert 1100 <= d[2].min() and d[2].max() <= 2900
    n0 = 0.07142864686293911
    coef0 = np.sqrt((3 * np.sin(th)) ** 2 + np.cos(th) ** 2)
    px0 = n0 * coef0 * np.cos(th) * (1 + (-np.sin(th) ** 2) * 8 / coef0 ** 2)
    assert np.max(np.abs(fin[6] - px0)) / n0 < 5e-4
    o = O.trazar(oracle_fields("anisotropy"), 11, 3, rb.DELTA_S, ms, lim, -2.0, -2.0, th[sub], record_stride=16, rec_rows=rows, nthreads=16)
    assert np.array_equal(d[:, sub], o["d_ray"]) and np.array_equal(fin[:, sub], o["final"])
    assert got.shape == o["s_ray"].shape and np.array_equal(got, o["s_ray"])


def test_cfg3_fisheye_full_1m_rays_every_row(rb, gpu_fields, oracle_fields):
    """cfg3 whole with the reference's full record: fisheye, 1 048 576 rays from (1, 0), op6, the calibrated DELTA_S = 2 pi / 303,
    s_ray[3040][6][R] = 153 GB on ONE MI355X, the library's default schedule.  Every recorded row of every 512th ray (2 048
    rays x 3 040 rows) against the oracle at 1e-9 per quantity, rows past a ray's last one zero (np.zeros, RT_bench.py:802), step
    counts equal, the device counter against sum(d_ray[2])."""
    from oracle import rt_oracle as O
    R = 1 << 20
    th = np.linspace(np.pi / 4, 3 * np.pi / 4, R)
    lim = LIMITS["fisheye"]
    step, ms = 2 * np.pi / 303, rb.N * 304
    b = rb.Batch(gpu_fields("fisheye"), 6, step, ms, lim, 1, th, 1.0, 0.0, record_stride=1, keep_n_ray=False)
    b.run()
    d, st = b.d_ray(), b.stats()
    sub = slice(0, R, 512)
    got = b.device_tensors()["s_ray"][:, :, sub].cpu().numpy()
    b.close()
    assert st["ray_steps"] == int(d[2].sum()) and st["live_rays"] == 0 and got.shape == (ms, 6, 2048)
    o = O.trazar(oracle_fields("fisheye"), 6, 1, step, ms, lim, 1.0, 0.0, th[sub], record_stride=1, nthreads=16)
    assert np.array_equal(d[2][sub], o["d_ray"][2])
    err = relerr(got, o["s_ray"])
    print(f"cfg3 full record: {got.shape[0]} rows x 6 x {got.shape[2]} rays vs oracle: max rel err {err:.2e}")
    assert err < REL and relerr(d[:2, sub], o["d_ray"][:2]) < REL
    last = d[2][sub].astype(int)
    for k in range(got.shape[2]):
        assert not got[last[k] + 1:, :, k].any()


def test_cfg4_fp32_full_8m_rays(rb, gpu_fields, oracle_fields):
    """cfg4 whole: vert_heterogeneous, 8 388 608 rays, fp32 field + step arithmetic on ONE MI355X (the config shards it
    over 8).  The reference is fp64-only, so the tolerances are measured ones, stated here -- against the ORACLE (fp64, the
    reference's algorithm) on every 512th ray (16 384 rays): the step count equal on at least 99 % of them and never more
    than one row apart (a ray whose last point lies within fp32's reach of the box's rim may leave a row earlier or later);
    on the rays with equal counts the end points within 2e-5, the traveltime within 2e-5 of its scale, both arclengths
    within 1e-5; p_x conserved over the whole fan."""
    from oracle import rt_oracle as O
    R = 1 << 23
    th = np.linspace(0, np.pi / 2, R)
    lim = LIMITS["vert_heterogeneous"]
    ms = int(np.ceil(80 / rb.DELTA_S) + 1)
    b = rb.Batch(gpu_fields("vert_heterogeneous", 1), 6, rb.DELTA_S, ms, lim, 1, th, -2.0, -2.0, record_stride=0)
    b.run()
    d32, f32, st = b.d_ray(), b.final(), b.stats()
    b.close()
    assert st["ray_steps"] == int(d32[2].sum()) and st["live_rays"] == 0
    sub = slice(0, R, 512)
    o = O.trazar(oracle_fields("vert_heterogeneous"), 6, 1, rb.DELTA_S, ms, lim, -2.0, -2.0, th[sub], record_stride=0, nthreads=16)
    dstep = np.abs(d32[2][sub] - o["d_ray"][2])
    same = dstep == 0
    err_xy = np.abs(f32[:2, sub] - o["final"][:2])[:, same].max()
    err_T = np.abs(f32[8, sub] - o["final"][8])[same].max() / np.abs(o["final"][8]).max()
    err_s = np.abs(d32[:2, sub] - o["d_ray"][:2])[:, same].max() / np.abs(o["d_ray"][:2]).max()
    print(f"cfg4 full, fp32 vs the fp64 oracle on {len(same)} rays: step count equal on {same.mean():.4%} (largest difference {int(dstep.max())} row); "
          f"on those: end points {err_xy:.2e}, traveltime {err_T:.2e} of its scale, arclengths {err_s:.2e} of theirs")
    assert dstep.max() <= 1 and same.mean() >= 0.99
    assert err_xy < 2e-5 and err_T < 2e-5 and err_s < 1e-5
    n0 = 0.07142864686293911
    assert np.max(np.abs(f32[6] - n0 * np.cos(th))) / n0 < 6e-4


# ------------------------------------------------------------------ SURVEY 8f rank 4, second stage: wavefronts across rays
@pytest.mark.parametrize("fixture,m,gam", [("wavefronts_vert_op6", 6, 1), ("wavefronts_aniso_op11", 11, 3)])
def test_wavefronts_vs_scipy_on_reference_trajectories(fixture, m, gam, rb, gpu_fields):
    """rtmi_wavefronts (device: sort by y, PCHIP across rays, derivative, normal angles, 100-point curve) against the
    reference's own code path (RT_bench.py:1005-1026, 1043-1044: np.argsort + scipy PchipInterpolator + .derivative())
    applied to the reference's trajectories -- 11 traveltimes x 31 rays of vert_heterogeneous op6 and of the anisotropy
    scenario (gamma = 3, op11), where the ray direction is NOT the wavefront normal (the reference draws both, :1016-1039)."""
    g = golden(fixture)
    b = rb.Batch(gpu_fields("vert_heterogeneous"), m, float(g["step"]), int(g["max_size"]), g["box"], gam, g["theta"], -2.0,
                 -2.0, record_stride=1)
    b.run()
    wf = b.wavefronts(g["times"])
    iso = b.isochrones(g["times"])
    b.close()
    assert len(wf) == 11
    seen = 0
    for it, w in enumerate(wf):
        n = int(g[f"count{it}"])
        assert w["count"] == n
        if n <= 1:
            assert len(w["dxdy"]) == 0 and len(w["x_fine"]) == 0      # the reference skips such wavefronts (:1011)
            continue
        seen += 1
        assert np.array_equal(w["ray"], g[f"ray{it}"])                # same order as np.argsort gives
        assert np.all(np.diff(w["y"]) > 0)
        for key in ("y", "x", "dxdy", "normal", "x_fine", "y_fine"):
            assert np.abs(w[key] - g[key + str(it)]).max() < 1e-10, (it, key)
        # |ray angle - normal angle|: the reference zips the ray-ordered angles with the y-sorted normals (:1032); the two
        # orders coincide on its fans, and the device pairs each point with its own angle
        assert np.abs(w["angle"] - iso[it, 2, w["ray"]]).max() == 0
        assert np.abs(w["angle_diff"] - np.abs(w["angle"] - w["normal"])).max() < 1e-15
        assert np.abs(w["angle_diff"] - g[f"angle_diff_ref{it}"]).max() < 1e-10
        if gam == 1:
            assert w["angle_diff"].max() < 0.05                       # isotropic: rays are normal to wavefronts (31 coarse points)
        else:
            assert w["angle_diff"].max() > 0.1                        # anisotropic: group and phase directions differ
    assert seen >= 8


def test_wavefronts_movie_frames_in_one_call(rb, gpu_fields, monkeypatch):
    """The reference's animation (RT_bench.py:1066-1102): 45 frames, travel_time = 0.01 + 0.01 frame, per frame the isochrone
    points sorted by y, PchipInterpolator x(y), 100 points of the curve.  One rtmi_wavefronts call makes all 45 (one sort of all
    points, regrouped per frame); compared with scipy doing what the reference's update(frame) does on the same isochrone
    points, with one call per frame, and with the chunked pass (RTMI_WF_CHUNK) -- the same numbers each way."""
    from scipy.interpolate import PchipInterpolator
    g = golden("wavefronts_vert_op6")
    b = rb.Batch(gpu_fields("vert_heterogeneous"), 6, float(g["step"]), int(g["max_size"]), g["box"], 1, g["theta"], -2.0, -2.0,
                 record_stride=1)
    b.run()
    times = 0.01 + 0.01 * np.arange(45)
    wf = b.wavefronts(times)
    iso = b.isochrones(times)
    single = [b.wavefronts([t])[0] for t in times[::7]]
    monkeypatch.setenv("RTMI_WF_CHUNK", "7")
    chunked = b.wavefronts(times)
    monkeypatch.delenv("RTMI_WF_CHUNK")
    b.close()
    assert len(wf) == 45
    drawn = 0
    for it, w in enumerate(wf):
        ok = ~np.isnan(iso[it, 0])
        assert w["count"] == ok.sum()
        for key in ("y", "x", "dxdy", "normal", "angle_diff", "x_fine", "y_fine", "ray"):
            assert np.array_equal(w[key], chunked[it][key], equal_nan=True), (it, key)
        if it % 7 == 0:
            for key in ("y", "x", "dxdy", "x_fine", "y_fine", "ray"):
                assert np.array_equal(w[key], single[it // 7][key], equal_nan=True), (it, key)
        if w["count"] < 2:
            continue
        xy = np.stack((iso[it, 0, ok], iso[it, 1, ok]), 1)                  # valid_ray_coord (:1082)
        srt = xy[np.argsort(xy[:, 1])]                                       # (:1092-1093)
        y_fine = np.linspace(srt[:, 1].min(), srt[:, 1].max(), 100)          # (:1096)
        x_fine = PchipInterpolator(srt[:, 1], srt[:, 0])(y_fine)             # (:1094, :1097)
        assert np.abs(w["y_fine"] - y_fine).max() < 1e-13 and np.abs(w["x_fine"] - x_fine).max() < 1e-10, it
        drawn += 1
    assert drawn >= 40


def test_wavefronts_large_fan_properties(rb, gpu_fields):
    """The same stage on a 65 536-ray fan with the rays in shuffled order (the sort has real work to do): y strictly
    increasing, ray indices a permutation of the rays that reach the traveltime, rays normal to the wavefront."""
    R = 65536
    rng = np.random.default_rng(3)
    th = rng.permutation(np.linspace(0.05, np.pi / 2 - 0.05, R))
    lim = LIMITS["vert_heterogeneous"]
    b = rb.Batch(gpu_fields("vert_heterogeneous"), 6, rb.DELTA_S, 30228, lim, 1, th, -2.0, -2.0, record_stride=1, rec_rows=3072,
                 sort_rays=True, keep_n_ray=False)
    b.run()
    times = [0.1, 0.2]
    wf = b.wavefronts(times, nfine=1000)
    iso = b.isochrones(times)
    b.close()
    for it, w in enumerate(wf):
        reach = ~np.isnan(iso[it, 0])
        assert w["count"] == reach.sum() and w["count"] > R // 2
        assert np.array_equal(np.sort(w["ray"]), np.nonzero(reach)[0])
        assert np.all(np.diff(w["y"]) > 0)
        assert np.array_equal(w["x"], iso[it, 0, w["ray"]]) and np.array_equal(w["y"], iso[it, 1, w["ray"]])
        assert np.median(w["angle_diff"]) < 1e-3 and w["angle_diff"][100:-100].max() < 2e-2
        assert np.all(np.diff(w["y_fine"]) > 0) and w["y_fine"][0] == w["y"][0] and w["y_fine"][-1] == w["y"][-1]
        assert np.abs(np.interp(w["y_fine"], w["y"], w["x"]) - w["x_fine"]).max() < 1e-6


@pytest.mark.parametrize("scen", ["vert_heterogeneous", "fisheye", "interface"])
def test_cell_polynomial_lookup_on_the_device(scen, rb, gpu_fields, oracle_fields, tmp_path_factory):
    """The fast-form methods' lookup (rtmi_debug_field_lookup: one polynomial per grid cell, rt_polytab.h) on the device:
      * bit for bit the host restatement (tests/native/polytab_check.cpp: the table built by the library's own host
        functions from the oracle's coefficient arrays -- which are the device's bits -- and looked up in the same operation
        order): the device's table IS that table, in every cell, rim cells and clamped points included;
      * within 2e-15 of the field's scale of FITPACK's own evaluation (the oracle's n_gradient = the reference's bits)."""
    import ctypes as C
    import subprocess
    from test_polytab_host import _table, _eval, ROOT, _dp
    so = str(tmp_path_factory.mktemp("polytab_gpu") / "libpolytab_check.so")
    subprocess.check_call(["g++", "-O2", "-ffp-contract=off", "-std=c++17", "-shared", "-fPIC", "-o", so,
                           os.path.join(ROOT, "tests", "native", "polytab_check.cpp")])
    L = C.CDLL(so)
    L.polytab_build.argtypes = [_dp, C.c_int, _dp, C.c_int, _dp, _dp, _dp, C.c_double, C.c_double, _dp]
    L.polytab_eval.argtypes = [_dp, C.c_int, C.c_int] + [C.c_double] * 6 + [C.c_long, _dp, _dp, _dp, _dp, _dp]
    F, OF = gpu_fields(scen), oracle_fields(scen)
    x, y, Z, cdy, cdx = OF.arrays()
    for a, b in zip(F.arrays()[2:], (Z, cdy, cdx)):
        assert np.array_equal(a, b)                   # the device's coefficient arrays are the oracle's bits
    tab, ihx, ihy = _table(L, x, y, Z, cdy, cdx)
    rng = np.random.default_rng(17)
    N = 100_000
    px = np.concatenate([rng.uniform(x[0], x[-1], N), rng.uniform(x[0], x[3], N // 10), rng.uniform(x[-4], x[-1], N // 10),
                         rng.uniform(x[0] - 1, x[-1] + 1, N // 10), x[[0, 1, 2, -3, -2, -1]]])
    py = np.concatenate([rng.uniform(y[0], y[-1], N), rng.uniform(y[0], y[-1], N // 10), rng.uniform(y[-4], y[-1], N // 10),
                         rng.uniform(y[0] - 1, y[-1] + 1, N // 10), y[[0, 1, 2, -3, -2, -1]]])
    dev = F.lookup_fast(px, py)
    host = _eval(L, tab, x, y, ihx, ihy, px, py)
    for name, a, b in zip(("n", "dn/dx", "dn/dy"), dev, host):
        assert np.array_equal(a, b), name
    want = OF.n_gradient(px, py)
    gscale = max(np.abs(cdx).max(), np.abs(cdy).max())
    for a, b, scale in zip(dev, want, (np.abs(Z).max(), gscale, gscale)):
        assert np.abs(a - b).max() < 2e-15 * scale


@pytest.mark.parametrize("scen,m,stride", [("vert_heterogeneous", 6, 1), ("fisheye", 7, 4), ("anisotropy", 11, 0)])
def test_step_repeat_graph_equals_single_steps(scen, m, stride, rb, gpu_fields):
    """rtmi_step_repeat (count launches of nsteps steps as one hipGraph) against the same launches one by one, and against one
    launch to termination: the same bits -- rows, d_ray, final state; the graph is rebuilt when nsteps / count change and
    survives a reset."""
    R = 700 if m != 11 else 150
    lim = LIMITS[scen]
    if scen == "fisheye":
        th, x0, y0, step, ms = np.linspace(np.pi / 4, 3 * np.pi / 4, R), 1.0, 0.0, 2 * np.pi / 303, 3040
    else:
        th, x0, y0, step, ms = np.linspace(0.06, np.pi / 2, R), -2.0, -2.0, rb.DELTA_S, 4000
    gam = 3 if scen == "anisotropy" else 1

    def make():
        return rb.Batch(gpu_fields(scen), m, step, ms, lim, gam, th, x0, y0, record_stride=stride)
    ref = make(); ref.run()
    want = (ref.d_ray(), ref.final(), ref.rows() if stride else None)
    ref.close()
    b = make()
    for nsteps, count in ((1, 37), (3, 11), (1, 37), (16, 300)):           # 37 + 33 + 37 + 4800 steps >= max_size
        b.step(nsteps, count)
    assert b.stats()["live_rays"] == 0 and b.stats()["launches"] == 37 + 11 + 37 + 300
    got = (b.d_ray(), b.final(), b.rows() if stride else None)
    for u, v in zip(got, want):
        assert (u is None and v is None) or np.array_equal(u, v)
    b.reset()
    while b.stats()["live_rays"]:
        b.step(16, 64)                                                     # the graph built above is reused after a reset
    assert np.array_equal(b.final(), want[1])
    c = make()
    for _ in range(37):
        c.step(1)
    c.step(3, 11)
    d = make(); d.step(1, 37); d.step(3, 11)
    assert np.array_equal(c.final(), d.final()) and np.array_equal(c.d_ray(), d.d_ray())
    b.close(); c.close(); d.close()


@pytest.mark.parametrize("qx,qy", [(8, 8), (9, 31), (64, 11), (120, 75)])
def test_cell_polynomials_on_arbitrary_grids(qx, qy, rb):
    """The polynomial table on caller-provided samples (rtmi_field_from_samples): the smallest grid the cubic fit takes (8 x 8:
    every cell is a rim cell), long thin ones, a larger one; a smooth field with a kink.  The fast-form lookup against
    FITPACK's arithmetic on the same device-built coefficients (rtmi_field_eval), and op1/2/6/8 trajectories of rays launched
    anywhere in the box -- through the not-a-knot end cells, some leaving the grid -- against the oracle."""
    from oracle import rt_oracle as O
    rng = np.random.default_rng(1000 * qx + qy)
    x = np.linspace(-1.0, 2.0, qx); y = np.linspace(0.5, 2.5, qy)
    X, Y = np.meshgrid(x, y)
    Z = 1.3 + 0.25 * np.sin(1.1 * X + 0.3) * np.cos(0.9 * Y) + 0.1 * np.abs(X - 0.4)
    delta = 0.5 * ((x[1] - x[0]) + (y[1] - y[0]))
    F = rb.Field.from_samples(x, y, Z, delta)
    OF = O.Field.from_samples(x, y, Z, delta)
    px = rng.uniform(x[0] - 0.3, x[-1] + 0.3, 20_000); py = rng.uniform(y[0] - 0.3, y[-1] + 0.3, 20_000)
    fast, fit = F.lookup_fast(px, py), F.n_gradient(px, py)
    scale = [np.abs(Z).max(), max(np.abs(F.arrays()[3]).max(), np.abs(F.arrays()[4]).max())]
    for a, b, sc in zip(fast, fit, (scale[0], scale[1], scale[1])):
        assert np.abs(a - b).max() < 5e-15 * sc
    R = 192
    x0 = rng.uniform(x[0] + 0.05, x[-1] - 0.05, R); y0 = rng.uniform(y[0] + 0.05, y[-1] - 0.05, R); th = rng.uniform(-np.pi, np.pi, R)
    lim = (x[0] + 0.01, x[-1] - 0.01, y[0] + 0.01, y[-1] - 0.01)
    for m in (1, 2, 6, 8):
        b = rb.Batch(F, m, 0.004, 1500, lim, 1, th, x0, y0, record_stride=0)
        b.run()
        d, fin = b.d_ray(), b.final()
        b.close()
        o = O.trazar(OF, m, 1, 0.004, 1500, lim, x0, y0, th, record_stride=0)
        same = d[2] == o["d_ray"][2]
        assert same.mean() > 0.98                                       # a ray grazing the box may leave one step apart
        assert relerr(fin[:, same], o["final"][:, same]) < 1e-9, m
    F.close()


def test_stats_totals_survive_reset(rb, gpu_fields):
    """rtmi_stats.kernel_ms_total / launches_total: the advance kernels of every pass since create (bench.py's kernel time per
    pass), while kernel_ms / launches restart with rtmi_batch_reset."""
    th = np.linspace(0.1, 1.4, 3000)
    b = rb.Batch(gpu_fields("vert_heterogeneous"), 6, rb.DELTA_S, 30228, LIMITS["vert_heterogeneous"], 1, th, -2.0, -2.0,
                 record_stride=0)
    per_pass = []
    for _ in range(3):
        b.reset(); b.run()
        per_pass.append(b.stats()["kernel_ms"])
    b.reset(); b.run(); b.reset(); b.run()                 # two passes without a stats call in between
    st = b.stats()
    assert st["launches"] == 1 and st["launches_total"] == 5
    assert st["kernel_ms"] > 0 and abs(st["kernel_ms_total"] - (sum(per_pass) + 2 * st["kernel_ms"])) < 0.5 * st["kernel_ms_total"]
    assert st["kernel_ms_total"] > sum(per_pass) + st["kernel_ms"]
    b.close()

