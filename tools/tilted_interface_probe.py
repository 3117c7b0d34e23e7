#!/usr/bin/env python3
"""Does the critical-ray hand-over depend on the interface scenario's axis-aligned wall?  The same sigmoid wall,
tilted and bent, given to both sides as SAMPLES (rtmi_field_from_samples / the oracle's from_samples): per field and method
the split of the 1 048 576-ray fan is found on the device, and the 4 096 contiguous rays around it are compared with the
oracle, every 16th row -- default (re-trace on) and no_retrace.  Checker run (tests/ material)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from raytracing_amd import rt_bench as rb          # noqa: E402
from oracle import rt_oracle as O                   # noqa: E402

lim = (-2, 20, -2, 4)
R = 1 << 20
W = int(os.environ.get("WINDOW", 4096))
th = np.linspace(2 * np.pi / 60, np.pi / 2, R)
ms = int(np.ceil(80 / rb.DELTA_S) + 1)
threads = min(O.max_threads(), os.cpu_count() or 1)
THICK = 0.005


def wall(d):
    return np.sqrt(2.0) - (np.sqrt(2.0) - 1.0) / (1.0 + np.exp(-np.clip(d / THICK, -700, 700)))


def fields():
    base = O.Field("interface", lim, rb.DELTA)
    x, y = base.arrays()[:2]
    X, Y = np.meshgrid(x, y)
    for deg in (3.0, 11.0):
        a = np.radians(deg)
        yield f"tilted {deg:g} deg", x, y, wall(-np.sin(a) * (X + 2.0) + np.cos(a) * Y)
    yield "arc r=40", x, y, wall(np.hypot(X - 9.0, Y + 40.0) - 40.0)
    yield "arc r=-60", x, y, wall(60.0 - np.hypot(X - 9.0, Y - 61.0))


def per_ray(a, w):
    out = []
    for q in ((0, 1), (2, 3), (4,), (5,)):
        out.append(np.abs(a[:, list(q)] - w[:, list(q)]).max(axis=(0, 1)) / np.abs(w[:, list(q)]).max())
    return np.array(out).max(axis=0)


print(f"# the interface's wall tilted and bent, fields from samples; {R}-ray fan, {W} contiguous rays around the split, every 16th row")
print(f"{'field':16s} {'op':>3s} {'order':10s} {'split at':>9s} {'same steps':>10s} {'largest':>9s} {'> 1e-9':>7s} {'retraced':>8s} {'overflow':>8s} {'ms':>8s}")
METHODS = [int(v) for v in os.environ.get("METHODS", "6,1").split(",")]
for name, x, y, Z in fields():
    F = rb.Field.from_samples(x, y, Z, rb.DELTA)
    OF = O.Field.from_samples(x, y, Z, rb.DELTA)
    for m in METHODS:
        b = rb.Batch(F, m, rb.DELTA_S, ms, lim, 1, th[::64], -2.0, -2.0, record_stride=0)
        b.run()
        fin = b.final()
        full = b.stats()
        b.close()
        k = int(np.argmax(np.abs(np.diff(fin[1])) + np.abs(np.diff(fin[0]))))
        i0 = min(max(0, k * 64 + 32 - W // 2), R - W)
        win = slice(i0, i0 + W)
        kw = dict(record_stride=16, rec_rows=600)
        o = O.trazar(OF, m, 1, rb.DELTA_S, ms, lim, -2.0, -2.0, th[win], nthreads=threads, **kw)
        for order in ("default", "noretrace"):
            b = rb.Batch(F, m, rb.DELTA_S, ms, lim, 1, th[win], -2.0, -2.0, retrace=order != "noretrace", **kw)
            b.run()
            s, d = b.rows(), b.d_ray()
            stt = b.stats()
            b.close()
            same = d[2] == o["d_ray"][2]
            dev = per_ray(s[:, :, same], o["s_ray"][:, :, same])
            print(f"{name:16s} {m:3d} {order:10s} {np.degrees(th[k * 64 + 32]):9.4f} {int(same.sum()):10d} "
                  f"{dev.max():9.1e} {int((dev > 1e-9).sum()):7d} {stt['retraced']:8d} {stt['retrace_overflow']:8d} {stt['kernel_ms']:8.3f}", flush=True)
    F.close()
