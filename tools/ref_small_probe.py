#!/usr/bin/env python3
"""Kernel time of SMALL reference-order batches (what the re-trace of critical rays runs as): interface fan windows and a
vert_heterogeneous fan, per method, rtmi_params.reference_order = 1.  usage: ref_small_probe.py [rays ...]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from raytracing_amd import rt_bench as rb          # noqa: E402

sizes = [int(v) for v in sys.argv[1:]] or [64, 4096]
for scen, lim, th_of in (("interface", (-2, 20, -2, 4), lambda R: np.radians(45.07) + np.arange(R) * 1.4e-6),
                         ("vert_heterogeneous", (-2, 5, -2.5, 1), lambda R: np.linspace(0.3, 1.2, R))):
    F = rb.Field.build(scen, lim, rb.DELTA)
    ms = int(np.ceil(80 / rb.DELTA_S) + 1)
    for m in (6, 2, 1):
        for R in sizes:
            b = rb.Batch(F, m, rb.DELTA_S, ms, lim, 1, th_of(R), -2.0, -2.0, record_stride=0, reference_order=1, launch_mode="plain")
            t = []
            for _ in range(3):
                b.reset(); b.run(); t.append(b.stats()["kernel_ms"])
            st = b.stats()
            print(f"{scen:20s} op{m} ref-order {R:6d} rays: {min(t):8.3f} ms  ({st['ray_steps'] / R:7.1f} steps per ray, longest {int(b.d_ray()[2].max())}) "
                  f"{1e3 * min(t) / b.d_ray()[2].max():6.3f} us per step of the longest ray, vgprs {st['vgprs']}", flush=True)
            b.close()
    F.close()
