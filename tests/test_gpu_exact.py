"""GPU: the methods that run in the reference's own operation order (op3/4/5/9/10/11, fp64; rt_exact.h).

Their results hinge on last bits -- a flipped golden-section comparison moves a step's angle by up to 6e-8, the
curvature advancement divides a cancelled difference of sines by a curvature as small as 1.5e-8 -- so "close" is not
good enough there and these tests ask for the SAME BITS as the oracle (which restates the reference operation by
operation and is pinned to it by tests/test_oracle_golden.py), and for 1e-9 on every ray against the reference's own
trajectories."""
import numpy as np
import pytest

import os

from conftest import GOLDEN, LIMITS, golden, sub_rows, traj_fixtures, traj_inputs

pytestmark = pytest.mark.gpu
EXACT = (3, 4, 5, 9, 10, 11)     # always in the reference's operation order: bit-identical to the oracle


@pytest.fixture(scope="module")
def rb():
    from raytracing_amd import rt_bench
    return rt_bench


@pytest.fixture(scope="module")
def fields(rb, oracle_fields):
    """Device fields (rtmi_field_build: genZ + interpolacion on the device) next to oracle fields."""
    cache = {}

    def get(scen):
        key = "vert_heterogeneous" if scen == "anisotropy" else scen
        if key not in cache:
            cache[key] = (rb.Field.build(key, LIMITS[key], rb.DELTA), oracle_fields(key))
        return cache[key]
    yield get
    for F, _ in cache.values():
        F.close()


def test_device_sincos_is_libm_bit_for_bit(rb):
    """rt_libm.h on the device against this host's libm (numpy's float64 sin/cos are libm's): every range of the
    algorithm, its switch points, table nodes, multiples of pi/2, both signs."""
    rng = np.random.default_rng(7)
    parts = [rng.uniform(lo, hi, 400000) for lo, hi in ((0, 0.126), (0.126, 0.855469), (0.855469, 2.426265), (2.426265, 8),
                                                         (8, 200), (200, 1e5), (1e5, 105414350.0), (1e-9, 1e-7))]
    parts += [rng.uniform(c - 1e-4, c + 1e-4, 50000) for c in (0.126, 0.855469, 2.426265)]
    k = np.arange(-4000, 4001)
    parts += [k / 128.0, np.nextafter(k / 128.0, 1e9), np.nextafter(k / 128.0, -1e9), k * (np.pi / 2),
              np.nextafter(k * (np.pi / 2), 1e9)]
    x = np.concatenate(parts)
    x = np.concatenate([x, -x, [0.0, -0.0, 1e-300, 2.0 ** -27, 2.0 ** -26]])
    s, c = rb.device_sincos(x)
    hs, hc = np.sin(x), np.cos(x)
    assert np.array_equal(s.view(np.uint64), hs.view(np.uint64)), f"sin differs on {np.sum(s != hs)} of {x.size}"
    assert np.array_equal(c.view(np.uint64), hc.view(np.uint64)), f"cos differs on {np.sum(c != hc)} of {x.size}"


@pytest.mark.parametrize("scen", ["vert_heterogeneous", "fisheye", "interface"])
def test_field_is_the_references_bits(scen, rb, fields):
    """genZ + interpolacion on the device: the samples (interface: numpy's array exp = SVML's, restated in k_sample), the
    np.gradient stencil and FITPACK regrid's Givens QR (k_givens / k_fpback) give the ORACLE's arrays bit for bit -- all
    three grids, every element -- and the oracle's are the reference's (tests/test_oracle_golden.py); checked here directly
    too: the 8x8 blocks of Z and of both coefficient arrays that the fixtures hold from the reference, and
    n_gradient (rtmi_field_eval, FITPACK's fpbisp arithmetic) at the reference's 1 024 random points per grid."""
    F, OF = fields(scen)
    for a, b in zip(F.arrays(), OF.arrays()):
        assert np.array_equal(a, b)
    g = golden(f"field_{scen}")
    _, _, Z, cdy, cdx = F.arrays()
    qy, qx = Z.shape
    for name, arr in (("Z", Z), ("cdy", cdy), ("cdx", cdx)):
        for tag, blk in (("c00", arr[:8, :8]), ("c11", arr[-8:, -8:]), ("mid", arr[qy // 2:qy // 2 + 8, qx // 2:qx // 2 + 8])):
            assert np.array_equal(blk, g[f"{name}_{tag}"]), (name, tag)
    n, gx, gy = F.n_gradient(g["px"], g["py"])
    scale = max(np.abs(cdx).max(), np.abs(cdy).max())
    # rtmi_field_eval: FITPACK's operation order with Newton reciprocals for the knot differences (rt::axis_basis), a few ulp
    assert np.abs(n - g["n"]).max() <= 1e-15 * np.abs(Z).max()
    assert np.abs(gx - g["gx"]).max() <= 4e-15 * scale and np.abs(gy - g["gy"]).max() <= 4e-15 * scale


def _bits_equal(a, b):
    return np.array_equal(np.asarray(a).view(np.uint64), np.asarray(b).view(np.uint64))


@pytest.mark.parametrize("m", EXACT)
def test_single_step_is_the_oracles_bits(m, rb, fields):
    """One selected_func call on the reference's 64 random states per method (fixture step_methods): bit-identical
    to the oracle, hence within the oracle's own 2e-15 of the reference's outputs."""
    from oracle import rt_oracle as O
    g = golden("step_methods")
    F, OF = fields("vert_heterogeneous")
    st, hist, ref = g[f"st{m}"], g[f"hist{m}"], g[f"out{m}"]
    R = st.shape[0]
    gam = 3 if m >= 10 else 1
    b = rb.Batch(F, m, float(g["step"]), 1 << 20, (-1e300, 1e300, -1e300, 1e300), gam, st[:, 2], st[:, 0], st[:, 1],
                 record_stride=0)
    state9 = np.zeros((9, R)); state9[:6] = st[:, :6].T
    b.set_state(state9, None, np.full(R, 3, dtype=np.int32))
    b.step(1)
    out = b.final()[:6].T
    b.close()
    o = O.single_step(OF, m, gam, float(g["step"]), st, hist)
    assert _bits_equal(out, o), f"op{m}: {np.sum(out != o)} of {out.size} values differ from the oracle"
    err = np.abs(out - ref) / np.maximum(np.abs(ref), 1e-3)
    assert err.max() < 1e-12


CASES = [("vert_heterogeneous", m, 31) for m in EXACT if m < 10] + [("anisotropy", 10, 31), ("anisotropy", 11, 31)] + \
        [("fisheye", m, 9) for m in (3, 4, 5, 9)] + [("interface", m, 16) for m in (3, 4, 5, 9)]
# reference fixtures of the same fans (tests/golden/traj_*): where one exists the device is ALSO compared with it
FIXTURE_OF = {("vert_heterogeneous", 31): "vert_op{m}", ("anisotropy", 31): "aniso_op{m}", ("fisheye", 9): "fisheye_op{m}_fan9",
              ("interface", 16): "interface_op{m}_16"}


@pytest.mark.parametrize("scen,m,R", CASES)
def test_trajectories_are_the_oracles_bits(scen, m, R, rb, fields):
    """Whole trajectories (every recorded row of every ray), d_ray and the final state: bit-identical to the oracle
    on the preset fans of all four scenarios, through both gather paths and the lane-refill kernel."""
    from oracle import rt_oracle as O
    F, OF = fields(scen)
    lim = LIMITS[scen]
    gam = 3 if scen == "anisotropy" else 1
    if scen == "fisheye":
        th, x0, y0, step, ms = np.linspace(np.pi / 4, 3 * np.pi / 4, R), 1.0, 0.0, 2 * np.pi / 303, 3040
    elif scen == "interface":
        th, x0, y0, step, ms = np.linspace(2 * np.pi / 60, np.pi / 2, R + 1)[:R], -2.0, -2.0, rb.DELTA_S, 30228
    else:
        th, x0, y0, step, ms = np.linspace(0, np.pi / 2, R), -2.0, -2.0, rb.DELTA_S, 30228
    rows = 9000
    o = O.trazar(OF, m, gam, step, ms, lim, x0, y0, th, record_stride=1, rec_rows=rows, want_n_ray=True, nthreads=8)
    # ... and the oracle's bits ARE the reference's on these fans (op3/4/5/9: every recorded row equal; op10/11: 3e-17)
    fx = os.path.join(GOLDEN, "traj_" + FIXTURE_OF[(scen, R)].format(m=m) + ".npz")
    if os.path.exists(fx):
        t = np.load(fx)
        strided, last = sub_rows(o["s_ray"], o["d_ray"], int(t["stride"]))
        nrow = min(strided.shape[0], t["strided"].shape[0])
        assert np.abs(strided[:nrow] - t["strided"][:nrow]).max() <= 1e-15 and np.abs(last - t["last"]).max() <= 1e-15
        if m < 10:
            assert np.array_equal(strided[:nrow], t["strided"][:nrow]) and np.array_equal(last, t["last"])
    for path, mode in ((1, "plain"), (2, "plain"), (2, "refill")):
        b = rb.Batch(F, m, step, ms, lim, gam, th, x0, y0, record_stride=1, rec_rows=rows, field_path=path, launch_mode=mode)
        b.run()
        d, fin = b.d_ray(), b.final()
        s, n = b.rows(want_n_ray=True)
        b.close()
        assert np.array_equal(d[2], o["d_ray"][2])
        assert _bits_equal(d, o["d_ray"]) and _bits_equal(fin, o["final"])
        assert _bits_equal(s, o["s_ray"]) and _bits_equal(n, o["n_ray"])


REF_ORDER_CASES = [(scen, m) for scen in ("vert_heterogeneous", "fisheye", "interface") for m in (1, 2, 6, 7, 8)]


@pytest.mark.parametrize("scen,m", REF_ORDER_CASES)
def test_reference_order_mode(scen, m, rb, fields):
    """rtmi_params.reference_order = 1: op1/2/6/7/8 too step in the reference's own operation order (rt_exact.h), with numpy's
    arctan2 (SVML) restated for op1/7/8.  Through the plain, the refill and the time-sliced kernel: the ORACLE's bits -- every
    recorded row, d_ray, final state -- and the REFERENCE's on its own fixture of the same fan (vert 31 rays, fisheye 9-ray
    fan, interface 16 rays): equal for 13 of the 15 fixtures, <= 2.3e-16 on the other two (x*x for numpy's scalar pow)."""
    from oracle import rt_oracle as O
    F, OF = fields(scen)
    name = {"vert_heterogeneous": f"traj_vert_op{m}", "fisheye": f"traj_fisheye_op{m}_fan9", "interface": f"traj_interface_op{m}_16"}[scen]
    t = golden(name)
    x0, y0, th = traj_inputs(t, scen)
    step, ms, lim = float(t["step"]), int(t["max_size"]), t["box"]
    o = O.trazar(OF, m, 1, step, ms, lim, x0, y0, th, record_stride=1, nthreads=8)
    for mode in ("plain", "refill", "sliced"):
        b = rb.Batch(F, m, step, ms, lim, 1, th, x0, y0, record_stride=1, reference_order=True, launch_mode=mode, slice_steps=300)
        b.run()
        d, fin, s = b.d_ray(), b.final(), b.rows()
        b.close()
        assert np.array_equal(d[2], t["d_ray"][2])
        assert _bits_equal(d, o["d_ray"]) and _bits_equal(fin, o["final"]) and _bits_equal(s, o["s_ray"]), mode
        strided, last = sub_rows(s, d, int(t["stride"]))
        err = max(np.abs(strided - t["strided"]).max(), np.abs(last - t["last"]).max())
        assert err < 1e-15, (mode, err)                      # the reference itself (0 except where x*x differs from pow)


@pytest.mark.parametrize("gam", [0.05, 0.3, 12.0, 50.0, 200.0])
@pytest.mark.parametrize("m", [10, 11])
def test_extreme_anisotropy_is_the_oracles_bits(m, gam, rb, fields):
    """The golden-section certificates of op10/op11 bound the search's third-order remainder with suprema of the momentum
    curve's derivatives, sampled per batch (gold_sup_derivatives): the curve's features are 1/max(gamma, 1/gamma) wide, so the
    sampling is refined with it (gamma 0.05 .. 50 here) and beyond 64 the certificates are off and every comparison runs the
    reference's arithmetic (gamma 200).  Every ray the oracle's bits either way."""
    from oracle import rt_oracle as O
    F, OF = fields("anisotropy")
    lim = LIMITS["anisotropy"]
    th = np.linspace(0, np.pi / 2, 64)
    ms = 400
    o = O.trazar(OF, m, gam, rb.DELTA_S, ms, lim, -2.0, -2.0, th, record_stride=0, nthreads=8)
    b = rb.Batch(F, m, rb.DELTA_S, ms, lim, gam, th, -2.0, -2.0, record_stride=0)
    b.run()
    d, fin = b.d_ray(), b.final()
    b.close()
    assert _bits_equal(d, o["d_ray"]) and _bits_equal(fin, o["final"]), (m, gam, np.abs(fin - o["final"]).max())


@pytest.mark.parametrize("scen", ["vert_heterogeneous", "fisheye", "interface"])
def test_op7_steps_in_reference_order_by_default(scen, rb, fields):
    """op7's new angle differentiates positions (RT_bench.py:370-372), so the positions' last bits matter: a default batch
    (rtmi_params.reference_order = 0) steps op7 in the reference's operation order throughout and gives the ORACLE's bits.  The
    two opt-ins: RTMI_ORDER_FAST_FIELD (the same step on the fused field lookup: every recorded row within 1e-10 of the oracle
    here, 8e-11 on 4 096-ray fans -- but 2.6e-9 on the critical ray of a 1 M-ray interface fan,
    profiles/r04_op7_offenders_interface_1m.txt) and RTMI_ORDER_FUSED (up to 8e-9 on interface rows)."""
    from bench import parity_relerr
    from oracle import rt_oracle as O
    F, OF = fields(scen)
    name = {"vert_heterogeneous": "traj_vert_op7", "fisheye": "traj_fisheye_op7_fan9", "interface": "traj_interface_op7_16"}[scen]
    t = golden(name)
    x0, y0, th = traj_inputs(t, scen)
    step, ms, lim = float(t["step"]), int(t["max_size"]), t["box"]
    o = O.trazar(OF, 7, 1, step, ms, lim, x0, y0, th, record_stride=1, nthreads=8)
    b = rb.Batch(F, 7, step, ms, lim, 1, th, x0, y0, record_stride=1)
    b.run()
    assert _bits_equal(b.d_ray(), o["d_ray"]) and _bits_equal(b.final(), o["final"]) and _bits_equal(b.rows(), o["s_ray"])
    b.close()
    errs = {}
    for order in ("fast_field", "fused"):
        b = rb.Batch(F, 7, step, ms, lim, 1, th, x0, y0, record_stride=1, reference_order=order)
        b.run()
        d, fin, s = b.d_ray(), b.final(), b.rows()
        b.close()
        assert np.array_equal(d[2], o["d_ray"][2])
        errs[order] = max(parity_relerr(s, o["s_ray"]), parity_relerr(fin, o["final"]), parity_relerr(d[:2], o["d_ray"][:2]))
    print(f"{scen} op7 vs oracle: fast_field {errs['fast_field']:.1e}, fused {errs['fused']:.1e}")
    assert errs["fast_field"] < 1e-10 and errs["fused"] < 1e-7


def test_reference_order_needs_fp64(rb):
    from raytracing_amd._lib import RtmiError
    F = rb.Field.build("vert_heterogeneous", LIMITS["vert_heterogeneous"], rb.DELTA, rb.F32)
    with pytest.raises(RtmiError, match="fp64"):
        rb.Batch(F, 6, rb.DELTA_S, 100, LIMITS["vert_heterogeneous"], 1, [0.3], -2.0, -2.0, reference_order=True)
    F.close()


@pytest.mark.parametrize("scen,m", [("vert_heterogeneous", 9), ("anisotropy", 11), ("fisheye", 5), ("vert_heterogeneous", 3),
                                    ("anisotropy", 10), ("fisheye", 9)])
def test_random_rays_are_the_oracles_bits(scen, m, rb, fields):
    """Seeded random launch points and directions anywhere on the padded grid with a coarse step: rays cross the
    not-a-knot end cells, sit on grid lines, and the golden-section cost has a large residual (many near-ties, so
    the exact re-evaluation of the filtered search is exercised constantly).  Still the oracle's bits."""
    from oracle import rt_oracle as O
    F, OF = fields(scen)
    rng = np.random.default_rng(300 + m)
    x, y, *_ = OF.arrays()
    R = 256
    gam = 3 if scen == "anisotropy" else 1
    x0 = rng.uniform(x[0] + 0.05, x[-1] - 0.05, R)
    y0 = rng.uniform(y[0] + 0.05, y[-1] - 0.05, R)
    x0[::3] = x[rng.integers(1, len(x) - 1, len(x0[::3]))]
    y0[1::3] = y[rng.integers(1, len(y) - 1, len(y0[1::3]))]
    th = rng.uniform(-np.pi, np.pi, R)
    lim = (x[0] + 0.01, x[-1] - 0.01, y[0] + 0.01, y[-1] - 0.01)
    step, ms = 0.011, 700
    b = rb.Batch(F, m, step, ms, lim, gam, th, x0, y0, record_stride=0)
    b.run()
    d, fin = b.d_ray(), b.final()
    b.close()
    o = O.trazar(OF, m, gam, step, ms, lim, x0, y0, th, record_stride=0, nthreads=8)
    assert _bits_equal(d, o["d_ray"]) and _bits_equal(fin, o["final"])


@pytest.mark.parametrize("name,scen,m", [t for t in traj_fixtures() if t[2] in (3, 4, 5, 9, 10, 11)])
def test_every_ray_within_1e9_of_the_reference(name, scen, m, rb, fields):
    """The north-star tolerance on EVERY ray (not a fraction of them) against the reference's own trajectories, golden-section
    and curvature methods, all scenarios, on device-built fields -- the product path end to end.  Measured: 0 on most,
    <= 3e-17 on all (the device gives the oracle's bits and the oracle the reference's) -- asserted at 1e-15."""
    t = golden("traj_" + name)
    F = fields(scen)[0]
    x0, y0, th = traj_inputs(t, scen)
    b = rb.Batch(F, m, float(t["step"]), int(t["max_size"]), t["box"], float(t["gamma"]), th, x0, y0, record_stride=1)
    b.run()
    d = b.d_ray(); s = b.rows()
    b.close()
    strided, last = sub_rows(s, d, int(t["stride"]))
    per_ray = np.max(np.abs(last - t["last"]) / np.maximum(np.abs(t["last"]), 1.0), axis=(0, 1))
    print(f"{name}: worst ray {per_ray.max():.2e}; same step count on {int(np.sum(d[2] == t['d_ray'][2]))}/{len(th)} rays")
    assert np.array_equal(d[2], t["d_ray"][2])
    assert per_ray.max() < 1e-15
    assert np.max(np.abs(strided - t["strided"]) / np.maximum(np.abs(t["strided"]), 1.0)) < 1e-15


def test_interface_curvature_conditioning(rb, fields, oracle_fields):
    """curvature_t (:361-363) forms [sin(th) - sin(th -+ curv*step)]/curv.  On the flat flanks of the interface sigmoid curv
    sits just above the straight-step threshold (1.5e-8), the subtraction cancels ~9 digits and the quotient turns a
    last-bit difference of the inputs into ~1e-8 of position per step.  Rounds 1-2 were 2e-7 from the reference on
    interface x op3/4/5 for that reason alone: their field differed from the reference's in last bits (libm exp and a banded
    LU against numpy's SVML exp and FITPACK's Givens QR), and op4's atan2 was libm's where numpy's is SVML's.  With those
    restated the oracle reproduces the reference's rows EXACTLY on all three methods, and the device the oracle's."""
    from oracle import rt_oracle as O
    for m in (3, 4, 5):
        t = golden(f"traj_interface_op{m}_16")
        x0, y0, th = traj_inputs(t, "interface")
        o = O.trazar(oracle_fields("interface"), m, 1, float(t["step"]), int(t["max_size"]), t["box"], x0, y0, th,
                     record_stride=1, rec_rows=9000, nthreads=8)
        _, last = sub_rows(o["s_ray"], o["d_ray"], int(t["stride"]))
        assert np.array_equal(last, t["last"]), m
