#!/usr/bin/env python3
"""The interface scenario's worst case for a form that is not the reference's bits: the rays of the 1 048 576-ray fan around
the CRITICAL angle (where the fan splits into reflected and refracted rays; a ray there runs along the interface and amplifies
any last-bit difference).  Finds the split on every 64th ray (device only), then compares the 4 096 CONTIGUOUS rays of the
full fan around it with the oracle, every 16th row, per quantity group, for each method's orders.  Checker run (tests/
material)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import parity_relerr                     # noqa: E402
from raytracing_amd import rt_bench as rb          # noqa: E402
from oracle import rt_oracle as O                   # noqa: E402

lim = (-2, 20, -2, 4)
R = 1 << 20
W = int(os.environ.get("WINDOW", 4096))
th = np.linspace(2 * np.pi / 60, np.pi / 2, R)
ms = int(np.ceil(80 / rb.DELTA_S) + 1)
F = rb.Field.build("interface", lim, rb.DELTA)
OF = O.Field("interface", lim, rb.DELTA)
threads = min(O.max_threads(), os.cpu_count() or 1)

b = rb.Batch(F, 6, rb.DELTA_S, ms, lim, 1, th[::64], -2.0, -2.0, record_stride=0)
b.run()
fin = b.final()
b.close()
jump = np.abs(np.diff(fin[1]))                      # final y: refracted rays leave through the top, reflected ones do not
k = int(np.argmax(jump))
i0 = max(0, k * 64 + 32 - W // 2)
win = slice(i0, i0 + W)
print(f"# the fan splits between rays {k * 64} and {k * 64 + 64} ({np.degrees(th[k * 64]):.5f} .. {np.degrees(th[k * 64 + 64]):.5f} deg); "
      f"window: rays {i0} .. {i0 + W - 1}, every 16th row, {threads} host threads")
print(f"{'op':>3s} {'order':10s} {'same steps':>10s} {'x y':>9s} {'p':>9s} {'T':>9s} {'theta':>9s} {'final':>9s} {'rays > 1e-9':>11s}")
for m in (1, 2, 6, 8, 7):
    o = O.trazar(OF, m, 1, rb.DELTA_S, ms, lim, -2.0, -2.0, th[win], record_stride=16, rec_rows=600, nthreads=threads)
    for order in (("default", "fast_field", "fused") if m == 7 else ("default", "reference")):
        b = rb.Batch(F, m, rb.DELTA_S, ms, lim, 1, th[win], -2.0, -2.0, record_stride=16, rec_rows=600, reference_order=order)
        b.run()
        s, d, fin = b.rows(), b.d_ray(), b.final()
        b.close()
        same = d[2] == o["d_ray"][2]
        cols = []
        over = np.zeros(W, bool)
        for q in ((0, 1), (2, 3), (4,), (5,)):
            a, w = s[:, list(q)][:, :, same], o["s_ray"][:, list(q)][:, :, same]
            scale = np.abs(w).max()
            e = np.abs(a - w).max(axis=(0, 1)) / scale
            cols.append(e.max())
            over[np.flatnonzero(same)[e > 1e-9]] = True
        ef = parity_relerr(fin[:, same], o["final"][:, same])
        print(f"{m:3d} {order:10s} {int(same.sum()):10d} " + " ".join(f"{c:9.1e}" for c in cols) + f" {ef:9.1e} {int(over.sum()):11d}", flush=True)
F.close()
