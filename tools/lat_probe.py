#!/usr/bin/env python3
"""Small-batch probe: 65 536 rays (one wave per SIMD) of the vert_heterogeneous fan over the whole quarter circle and over
1/16 of it (the angular spacing of the 1 M-ray fan: waves that sit in one cell), op6 fp64, no record; kernel ms per pass."""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from raytracing_amd import rt_bench as rb

F = rb.Field.build("vert_heterogeneous")
lim = (-2, 5, -2.5, 1)
ms = int(np.ceil(80 / rb.DELTA_S) + 1)
for R in (65536, 131072):
    for span in (1.0, 1.0 / 16):
        th = np.linspace(0.5 * (1 - span) * np.pi / 2, 0.5 * (1 + span) * np.pi / 2, R)
        b = rb.Batch(F, 6, rb.DELTA_S, ms, lim, 1, th, -2.0, -2.0, record_stride=0)
        t = []
        for it in range(6):
            b.reset(); b.run(); t.append(b.stats()["kernel_ms"])
        st = b.stats()
        print(f"R {R} span {span:6.4f}: kernel ms {min(t):.3f} (runs {' '.join('%.3f' % v for v in t)}) ray-steps {st['ray_steps']} vgpr {st['vgprs']} "
              f"steps of the longest ray {int(b.d_ray()[2].max())}")
        b.close()
