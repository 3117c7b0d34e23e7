#!/usr/bin/env python3
"""Where do op7's orders leave the oracle most, on the 1 048 576-ray interface fan (every 64th ray, every 16th row)?  Per quantity
group, the worst rays and rows.  Checker run (tests/ material).  The finding (profiles/r04_op7_offenders_interface_1m.txt): the
reference-order step on the fast field lookup (RTMI_ORDER_FAST_FIELD) leaves 1e-9 on the one ray at the interface's critical
angle; reference order throughout -- op7's default -- gives the oracle's bits."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from raytracing_amd import rt_bench as rb          # noqa: E402
from oracle import rt_oracle as O                   # noqa: E402

lim = (-2, 20, -2, 4)
R = 1 << 20
th = np.linspace(2 * np.pi / 60, np.pi / 2, R)
sub = slice(0, R, 64)
ms = int(np.ceil(80 / rb.DELTA_S) + 1)
F = rb.Field.build("interface", lim, rb.DELTA)
OF = O.Field("interface", lim, rb.DELTA)
o = O.trazar(OF, 7, 1, rb.DELTA_S, ms, lim, -2.0, -2.0, th[sub], record_stride=16, rec_rows=600, nthreads=min(O.max_threads(), os.cpu_count()))
for order in ("default", "fast_field", "fused"):
    b = rb.Batch(F, 7, rb.DELTA_S, ms, lim, 1, th[sub], -2.0, -2.0, record_stride=16, rec_rows=600, reference_order=order)
    b.run()
    s, d = b.rows(), b.d_ray()
    b.close()
    assert np.array_equal(d[2], o["d_ray"][2])
    print(f"--- order {order}")
    for name, q in (("x y", (0, 1)), ("p", (2, 3)), ("T", (4,)), ("theta", (5,))):
        a, w = s[:, list(q)], o["s_ray"][:, list(q)]
        err = np.abs(a - w)
        scale = np.abs(w).max()
        per_ray = err.max(axis=(0, 1)) / scale
        worst = np.argsort(-per_ray)[:4]
        print(f"{name:6s} scale {scale:.3g}  max {per_ray.max():.2e}  rays over 1e-9: {(per_ray > 1e-9).sum()} of {len(per_ray)}  over 1e-10: {(per_ray > 1e-10).sum()}"
              f"  worst rays (theta0 deg, last row, err): " + ", ".join(f"({np.degrees(th[sub][k]):.4f}, {int(d[2, k])}, {per_ray[k]:.1e})" for k in worst))
