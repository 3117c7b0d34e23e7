#!/usr/bin/env python3
"""Which per-step quantity predicts a ray's AMPLIFICATION (how far its rows move when the launch angle moves by 1e-12 of
itself)?  CPU only, on the oracle's trajectories, for the interface scenario's wall as it is and tilted / bent (fields from
samples, as tools/tilted_interface_probe.py builds them).  Candidates, each a sum over the ray's steps times DELTA_S:
  A  the cell's steepness (largest |Hessian n|_inf over its corners), counted when the ray heads within |sin| < 0.02 of the
     iso-lines (rt::hover_update up to round 5's first form)
  B  the same steepness times |nu . g| (nu the ray's normal, g the unit gradient): no angle threshold
  C  like B with the cell's steepness taken from the DEFOCUSING part only: max(0, g' H g) over the corners
  D  sqrt(max(0, nu' H nu) / n) with the Hessian at the ray's own position: the growth rate of the paraxial (Jacobi) equation
Printed per field: correlation with ln(amplification), and for the levels 1e3 / 3e3 / 1e4 the smallest sum among the rays
above the level and how many of the sampled rays a limit there would flag.  Checker-side tool (imports oracle/)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import rt_oracle as O                  # noqa: E402

DELTA, DELTA_S = 0.01, 0.001
lim = (-2, 20, -2, 4)
R = 1 << 20
W, STRIDE = int(os.environ.get("WINDOW", 768)), int(os.environ.get("STRIDE", 8))
ROWS = int(os.environ.get("ROWS", 12000))
M = int(os.environ.get("METHOD", 6))
th = np.linspace(2 * np.pi / 60, np.pi / 2, R)
ms = int(np.ceil(80 / DELTA_S) + 1)
threads = min(O.max_threads(), os.cpu_count() or 1)
THICK = 0.005


def wall(d):
    return np.sqrt(2.0) - (np.sqrt(2.0) - 1.0) / (1.0 + np.exp(-np.clip(d / THICK, -700, 700)))


base = O.Field("interface", lim, DELTA)
x, y = base.arrays()[:2]
X, Y = np.meshgrid(x, y)
qx, qy = len(x), len(y)
hx, hy = (x[-1] - x[0]) / (qx - 1), (y[-1] - y[0]) / (qy - 1)
lam0 = 40.0 / min(x[-1] - x[0], y[-1] - y[0])


def split_index(deg):
    return int(round((np.radians(deg) - th[0]) / (th[-1] - th[0]) * (R - 1)))


CASES = {                                           # name: (samples, the op6 split tools/tilted_interface_probe.py found on the device)
    "straight": (None, 487752),
    "tilted3": (wall(-np.sin(np.radians(3.0)) * (X + 2.0) + np.cos(np.radians(3.0)) * Y), split_index(48.1820)),
    "tilted11": (wall(-np.sin(np.radians(11.0)) * (X + 2.0) + np.cos(np.radians(11.0)) * Y), split_index(56.0673)),
    "arc-60": (wall(60.0 - np.hypot(X - 9.0, Y - 61.0)), split_index(38.7946)),
}


def hessian(F, px, py, e=2.5e-7):
    """[Hxx, Hxy, Hyx, Hyy], n, gx, gy at the points, by one-sided differences of the reference's gradient splines"""
    n0, gx0, gy0 = F.n_gradient(px, py)
    _, gx1, gy1 = F.n_gradient(px + e, py)
    _, gx2, gy2 = F.n_gradient(px, py + e)
    return np.array([(gx1 - gx0) / e, (gx2 - gx0) / e, (gy1 - gy0) / e, (gy2 - gy0) / e]), n0, gx0, gy0


def cell_steepness(F):
    """per cell [qy-1, qx-1]: (largest |H|_inf over the corners / smallest n)^(1/2), and the same from max(0, g' H g)"""
    e = 1e-6
    hin = np.zeros((qy - 1, qx - 1)); hpl = np.zeros_like(hin); nm = np.full_like(hin, np.inf)
    CX, CY = np.meshgrid(x[:-1], y[:-1])
    Zs = F.arrays()[2]
    pad = np.pad(Zs, 3, mode="edge")                 # only cells whose spline support (the 6 x 6 samples around) is not constant
    hi = np.max([pad[i:i + qy, j:j + qx] for i in range(7) for j in range(7)], axis=0)
    lo = np.min([pad[i:i + qy, j:j + qx] for i in range(7) for j in range(7)], axis=0)
    live = ((hi - lo) > 1e-7)[:-1, :-1]
    for cu in (0, 1):
        for cv in (0, 1):
            px = (CX + (e if cu == 0 else hx - 2 * e))[live]; py = (CY + (e if cv == 0 else hy - 2 * e))[live]
            H, n0, gx, gy = hessian(F, px, py)
            g2 = np.maximum(gx * gx + gy * gy, 1e-300)
            gHg = (H[0] * gx * gx + (H[1] + H[2]) * gx * gy + H[3] * gy * gy) / g2
            hin[live] = np.maximum(hin[live], np.maximum(np.abs(H[0]) + np.abs(H[1]), np.abs(H[2]) + np.abs(H[3])))
            hpl[live] = np.maximum(hpl[live], np.maximum(gHg, 0.0))
            nm[live] = np.minimum(nm[live], n0)
    nm[~live] = 1.0
    li, lp = np.sqrt(hin / nm), np.sqrt(hpl / nm)
    return np.where(li >= lam0, li, 0.0), np.where(lp >= lam0, lp, 0.0)


print(f"# op{M}; {W} rays, every {STRIDE}th of the {R}-ray fan, around each field's split; rows 0 .. {ROWS}; lambda_0 = {lam0:.3f}; {threads} host threads")
for name in os.environ.get("CASES", "straight,tilted11,arc-60").split(","):
    Z, split = CASES[name]
    F = base if Z is None else O.Field.from_samples(x, y, Z, DELTA)
    lin, lpl = cell_steepness(F)
    idx = np.arange(split - W // 2 * STRIDE, split + W // 2 * STRIDE, STRIDE)
    if os.environ.get("FAN"):                       # the whole fan, every FAN-th ray: where else do the sums grow?
        idx = np.arange(0, R, int(os.environ["FAN"]))
    kw = dict(nthreads=threads, record_stride=1, rec_rows=ROWS)
    o = O.trazar(F, M, 1, DELTA_S, ms, lim, -2.0, -2.0, th[idx], **kw)
    o1 = O.trazar(F, M, 1, DELTA_S, ms, lim, -2.0, -2.0, th[idx] * (1 + 1e-12), **kw)
    s, s1, last = o["s_ray"], o1["s_ray"], o["d_ray"][2].astype(int)
    moved = np.max([np.abs(s[:, q] - s1[:, q]).max(axis=(0, 1)) / np.abs(s[:, q]).max() for q in ([0, 1], [2, 3], [5])], axis=0)
    other = o1["d_ray"][2] != o["d_ray"][2]          # a different number of steps under the perturbation: rays at a rim, set aside
    amp = np.maximum(moved / 1e-12, 1.0)
    del s1, o1
    rows = s.shape[0]
    valid = (np.arange(rows)[:, None] <= np.minimum(last, rows - 1)[None, :])[:-1]
    px, py, t = s[:-1, 0, :], s[:-1, 1, :], s[:-1, 5, :]           # a step starts at row k with the tangent of row k
    jx = np.clip(np.floor((px - x[0]) / hx).astype(int), 0, qx - 2); jy = np.clip(np.floor((py - y[0]) / hy).astype(int), 0, qy - 2)
    li, lp = lin[jy, jx], lpl[jy, jx]
    near = (li > 0) & valid
    sums = {k: np.zeros(len(idx)) for k in "ABCD"}
    H, n0, gx, gy = hessian(F, px[near], py[near])
    ux, uy = np.cos(t[near]), np.sin(t[near])
    g2 = np.maximum(gx * gx + gy * gy, 1e-300)
    dot2 = (ux * gx + uy * gy) ** 2 / g2
    nug = np.sqrt(np.maximum(0.0, 1.0 - dot2))
    nHn = H[0] * uy * uy - (H[1] + H[2]) * ux * uy + H[3] * ux * ux          # nu = (-uy, ux)
    col = np.nonzero(near)[1]
    np.add.at(sums["A"], col, li[near] * (dot2 < 4e-4) * DELTA_S)
    np.add.at(sums["B"], col, li[near] * nug * DELTA_S)
    np.add.at(sums["C"], col, lp[near] * nug * DELTA_S)
    np.add.at(sums["D"], col, np.sqrt(np.maximum(nHn, 0.0) / n0) * DELTA_S)
    print(f"{name}: split at ray {split}; steep cells {int((lin > 0).sum())} (defocusing {int((lpl > 0).sum())}); amplification > 1e3 / 1e4 / 1e5: "
          f"{(amp > 1e3).sum() * STRIDE} / {(amp > 1e4).sum() * STRIDE} / {(amp > 1e5).sum() * STRIDE} rays of the million ({int(other.sum())} sampled rays change their step count); largest hover row {int((near * np.arange(rows - 1)[:, None]).max())}")
    for k in "ABCD":
        v = sums[k]
        row = []
        for A in (1e3, 3e3, 1e4):
            need = (amp > A) & ~other
            if not need.any():
                continue
            T = v[need].min()
            row.append(f"> {A:.0e}: limit {T:5.2f} flags {(v >= T).sum() * STRIDE:5d} (needed {need.sum() * STRIDE:4d})")
        if os.environ.get("FAN"):
            edges = np.linspace(0, len(idx), 17).astype(int)
            print(f"   {k} over the fan, largest per 1/16th: " + " ".join(f"{v[a:b].max():6.2f}" for a, b in zip(edges[:-1], edges[1:])))
        print(f"   {k}: corr with ln amp {np.corrcoef(v, np.log(np.minimum(amp, 1e12)))[0, 1]:5.2f} | " + " | ".join(row), flush=True)
    del s, o
