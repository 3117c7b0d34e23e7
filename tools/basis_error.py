#!/usr/bin/env python3
"""How far the uniform cubic B-spline polynomials are from FITPACK's fpbspl on the reference's true knots
(numpy.linspace grids of genZ, RT_bench.py:429).  Backs the bound quoted in rt_device.h (axis_eval)."""
import numpy as np


def fpbspl(t, x, l):
    h = [1.0, 0, 0, 0]
    for j in range(1, 4):
        hh = h[:j]; h[0] = 0.0
        for i in range(j):
            li = l + 1 + i; lj = li - j
            f = hh[i] / (t[li] - t[lj])
            h[i] = h[i] + f * (t[li] - x); h[i + 1] = f * (x - t[lj])
    return np.array(h)


for name, (a, b, q) in {"vert x": (-5.0, 8.0, 737), "vert y": (-5.5, 4.0, 539), "interface x": (-5.0, 23.0, 1587),
                        "interface y": (-5.0, 7.0, 681), "fisheye": (-4.5, 4.5, 511)}.items():
    x = np.linspace(a, b, q); t = np.concatenate(([a] * 4, x[2:-2], [b] * 4))
    rng = np.random.default_rng(0); worst = 0.0
    for _ in range(20000):
        j = int(rng.integers(5, q - 7)); v = x[j] + rng.random() * (x[j + 1] - x[j])
        if not (x[j] <= v < x[j + 1]):
            continue
        w = fpbspl(t, v, j + 2)
        f = 1.0 / (x[j + 1] - x[j]); om = f * (x[j + 1] - v); u = f * (v - x[j])
        wu = np.array([om * om * om / 6, u * u * (u * 0.5 - 1) + 2 / 3, om * om * (om * 0.5 - 1) + 2 / 3, u * u * u / 6])
        worst = max(worst, np.abs(w - wu).max())
    print(f"{name:12s} q={q:5d}  max |w_uniform - w_fpbspl| = {worst:.2e}")
