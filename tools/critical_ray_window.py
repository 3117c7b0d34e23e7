#!/usr/bin/env python3
"""The interface scenario's worst case for a form that is not the reference's bits: the rays of the 1 048 576-ray fan around
the CRITICAL angle (where the fan splits into reflected and refracted rays; a ray there runs along the interface and amplifies
any last-bit difference).  Per method: finds the split on every 64th ray (device only), then compares the 4 096 CONTIGUOUS
rays of the full fan around it with the oracle, every 16th row, per quantity group, for each of the method's orders -- and
measures the CONDITIONING of those rays: how far the oracle's own rows move when the launch angle moves by 1e-12 of itself
(a fused form's roundings differ from the reference's by ~1e-16 per step over the ~5 000 steps to the interface: 1e-14 .. 1e-13
by the time the ray arrives).  A ray whose own rows move by more than 1e-9 for such a change of its input cannot be
reproduced to 1e-9 by anything but the same roundings.  Checker run (tests/ material)."""
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
PERT = 1e-12



def per_ray(a, w):
    """largest difference of each ray's rows, per quantity group, relative to the group's largest magnitude in w: [4][R]"""
    out = []
    for q in ((0, 1), (2, 3), (4,), (5,)):
        out.append(np.abs(a[:, list(q)] - w[:, list(q)]).max(axis=(0, 1)) / np.abs(w[:, list(q)]).max())
    return np.array(out)


print(f"# interface scenario, {R} rays; windows of {W} contiguous rays around each method's split, every 16th row; {threads} host threads")
print("# columns x y / p / T / theta: largest difference from the oracle over the window, relative to the quantity's largest magnitude")
print(f"# '> 1e-9': rays with any quantity beyond 1e-9; 'ill': those of them whose ORACLE rows move more than that when theta_0 becomes theta_0 (1 + {PERT:g});")
print("# 'worst ratio': the largest (difference from the oracle) / (the oracle's own movement under that perturbation) over the rays beyond 1e-9")
print("# 'default': the library's default (critical rays re-traced in reference order by themselves, rtmi_params.no_retrace = 0); 'noretrace': the fused")
print("# forms alone (no_retrace = 1: what the default was until round 4); 'retraced': rays the default handed over (rtmi_stats.retraced)")
print(f"{'op':>3s} {'order':10s} {'split at':>9s} {'same steps':>10s} {'x y':>9s} {'p':>9s} {'T':>9s} {'theta':>9s} {'final':>9s} {'> 1e-9':>7s} {'ill':>5s} {'worst ratio':>11s} {'retraced':>8s} {'ms':>8s}")
METHODS = [int(v) for v in os.environ.get("METHODS", "1,2,6,8,7").split(",")]
for m in METHODS:
    b = rb.Batch(F, m, rb.DELTA_S, ms, lim, 1, th[::64], -2.0, -2.0, record_stride=0, reference_order="fused")
    b.run()
    fin = b.final()
    b.close()
    k = int(np.argmax(np.abs(np.diff(fin[1]))))     # final y: refracted rays leave through the top, reflected ones do not
    i0 = min(max(0, k * 64 + 32 - W // 2), R - W)
    win = slice(i0, i0 + W)
    kw = dict(record_stride=16, rec_rows=600)
    o = O.trazar(OF, m, 1, rb.DELTA_S, ms, lim, -2.0, -2.0, th[win], nthreads=threads, **kw)
    o1 = O.trazar(OF, m, 1, rb.DELTA_S, ms, lim, -2.0, -2.0, th[win] * (1 + PERT), nthreads=threads, **kw)
    moved = per_ray(o1["s_ray"], o["s_ray"]).max(axis=0)                 # the oracle's own movement under the perturbation, per ray
    moved[o1["d_ray"][2] != o["d_ray"][2]] = np.inf                      # (a different number of steps: any difference goes)
    for order in (("default", "fast_field", "fused") if m == 7 else ("default", "noretrace", "reference")):
        b = rb.Batch(F, m, rb.DELTA_S, ms, lim, 1, th[win], -2.0, -2.0, reference_order="default" if order == "noretrace" else order,
                     retrace=order != "noretrace", **kw)
        b.run()
        s, d, fin = b.rows(), b.d_ray(), b.final()
        stt = b.stats()
        b.close()
        same = d[2] == o["d_ray"][2]
        e = per_ray(s[:, :, same], o["s_ray"][:, :, same])
        dev = e.max(axis=0)
        over = dev > 1e-9
        ill = over & (moved[same] > dev)
        ratio = (dev[over] / np.maximum(moved[same][over], 1e-300)).max() if over.any() else 0.0
        ef = parity_relerr(fin[:, same], o["final"][:, same])
        print(f"{m:3d} {order:10s} {np.degrees(th[k * 64 + 32]):9.4f} {int(same.sum()):10d} " + " ".join(f"{c:9.1e}" for c in e.max(axis=1)) +
              f" {ef:9.1e} {int(over.sum()):7d} {int(ill.sum()):5d} {ratio:11.3f} {stt['retraced']:8d} {stt['kernel_ms']:8.3f}", flush=True)
        if stt["retrace_overflow"]:
            print(f"#      retrace_overflow {stt['retrace_overflow']}")
        for r in np.flatnonzero(over)[:8] if order != "fused" else ():
            print(f"#      ray {i0 + np.flatnonzero(same)[r]} ({np.degrees(th[i0 + np.flatnonzero(same)[r]]):.6f} deg): {dev[r]:.1e} from the oracle; the oracle itself moves {moved[same][r]:.1e}")
F.close()
