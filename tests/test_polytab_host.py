"""The per-cell polynomial form of the field (raytracing_amd/csrc/rt_polytab.h) checked on the CPU: the table built by the
library's own host functions (compiled here with g++) and looked up like rt::PolyGather does, against FITPACK's B-spline
arithmetic on the same coefficients (the oracle's n_gradient, which is the reference's bits).  No GPU involved."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from conftest import LIMITS

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_dp = C.POINTER(C.c_double)


def _p(a):
    return a.ctypes.data_as(_dp)


@pytest.fixture(scope="module")
def polylib(tmp_path_factory):
    so = str(tmp_path_factory.mktemp("polytab") / "libpolytab_check.so")
    subprocess.check_call(["g++", "-O2", "-ffp-contract=off", "-mfma", "-std=c++17", "-shared", "-fPIC", "-o", so,
                           os.path.join(ROOT, "tests", "native", "polytab_check.cpp")])
    L = C.CDLL(so)
    L.mdiv_mismatches.restype = C.c_long
    L.mdiv_mismatches.argtypes = [C.c_double, C.c_double, C.c_int, C.c_long, C.c_ulonglong]
    L.polytab_build.argtypes = [_dp, C.c_int, _dp, C.c_int, _dp, _dp, _dp, C.c_double, C.c_double, _dp]
    L.polytab_eval.argtypes = [_dp, C.c_int, C.c_int] + [C.c_double] * 6 + [C.c_long, _dp, _dp, _dp, _dp, _dp]
    return L


def _table(L, x, y, Z, cdy, cdx):
    qx, qy = len(x), len(y)
    hx = (x[-1] - x[0]) / (qx - 1); hy = (y[-1] - y[0]) / (qy - 1)
    tab = np.zeros((qy - 1) * (qx - 1) * 40)
    L.polytab_build(_p(x), qx, _p(y), qy, _p(Z), _p(cdx), _p(cdy), 1.0 / hx, 1.0 / hy, _p(tab))
    return tab, 1.0 / hx, 1.0 / hy


def _eval(L, tab, x, y, ihx, ihy, px, py):
    n = np.empty_like(px); gx = np.empty_like(px); gy = np.empty_like(px)
    L.polytab_eval(_p(tab), len(x), len(y), x[0], x[-1], ihx, y[0], y[-1], ihy, len(px), _p(px), _p(py), _p(n), _p(gx), _p(gy))
    return n, gx, gy


@pytest.mark.parametrize("scen", ["vert_heterogeneous", "fisheye", "interface"])
def test_cell_polynomials_equal_fitpack(scen, polylib, oracle_fields):
    """Random points over the whole grid -- interior, the double-width not-a-knot cells at the rim, points outside (clamped):
    the polynomial lookup agrees with FITPACK's evaluation to a few ulp of the field's scale."""
    F = oracle_fields(scen)
    x, y, Z, cdy, cdx = F.arrays()
    tab, ihx, ihy = _table(polylib, x, y, Z, cdy, cdx)
    rng = np.random.default_rng(5)
    N = 200_000
    px = np.concatenate([rng.uniform(x[0], x[-1], N), rng.uniform(x[0], x[3], N // 10), rng.uniform(x[-4], x[-1], N // 10),
                         rng.uniform(x[0] - 1, x[-1] + 1, N // 10), x[[0, 1, 2, -3, -2, -1]], x[[0, 5, -1]] + 1e-13])
    py = np.concatenate([rng.uniform(y[0], y[-1], N), rng.uniform(y[0], y[-1], N // 10), rng.uniform(y[-4], y[-1], N // 10),
                         rng.uniform(y[0] - 1, y[-1] + 1, N // 10), y[[0, 1, 2, -3, -2, -1]], y[[0, 5, -1]] - 1e-13])
    got = _eval(polylib, tab, x, y, ihx, ihy, px, py)
    want = F.n_gradient(px, py)
    gscale = max(np.abs(cdx).max(), np.abs(cdy).max())        # vert_heterogeneous: dn/dx is rounding noise around 0
    for name, a, b, scale in zip(("n", "dn/dx", "dn/dy"), got, want, (np.abs(Z).max(), gscale, gscale)):
        err = np.abs(a - b).max() / scale
        print(f"{scen} {name}: max |poly - fitpack| / max|coef| = {err:.2e}")
        assert err < 2e-15


def test_cell_polynomials_noisy_samples(polylib):
    """A rough field (noise on the samples: large third differences) on a small non-square grid, from_samples path."""
    from oracle import rt_oracle as O
    rng = np.random.default_rng(11)
    qx, qy = 23, 17
    x = np.linspace(-1.0, 2.0, qx); y = np.linspace(0.5, 2.5, qy)
    X, Y = np.meshgrid(x, y)
    Z = 1.2 + 0.3 * np.sin(1.3 * X) * np.cos(0.7 * Y) + 0.05 * rng.standard_normal(X.shape)
    F = O.Field.from_samples(x, y, Z, 0.11)
    x, y, Z, cdy, cdx = F.arrays()
    tab, ihx, ihy = _table(polylib, x, y, Z, cdy, cdx)
    px = rng.uniform(-1.5, 2.5, 50_000); py = rng.uniform(0.0, 3.0, 50_000)
    got = _eval(polylib, tab, x, y, ihx, ihy, px, py)
    want = F.n_gradient(px, py)
    for a, b, c in zip(got, want, (Z, cdx, cdy)):
        assert np.abs(a - b).max() < 4e-15 * np.abs(c).max()


def test_markstein_division_equals_ieee_division(polylib):
    """rt::ex::mdiv (the reference-order lookup's divisions by knot differences: q0 = a r, q = q0 + (a - d q0) r with
    r = RN(1/d) from the field build's table) against the IEEE quotient, on the three scenarios' axes: 0 mismatches on 3e8
    pairs.  (On the device every reference-order trajectory test holds the result to the oracle's bits, whose C divides.)"""
    for lo, hi, m, seed in ((-5.0, 8.0, 737, 1), (-5.5, 4.0, 539, 2), (-4.5, 4.5, 510, 3), (-5.0, 23.0, 1587, 4), (-5.0, 7.0, 680, 5)):
        assert polylib.mdiv_mismatches(lo, hi, m, 60_000_000, seed) == 0
