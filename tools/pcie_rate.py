#!/usr/bin/env python3
"""Host-boundary (PCIe-inclusive) cost of the 1M-ray bench batch: upload of launch conditions, read-back of
d_ray + final state, and of a block of trajectory rows (to extrapolate the full s_ray)."""
import sys, time
sys.path.insert(0, ".")
import numpy as np
from raytracing_amd import rt_bench as rb
R = 1 << 20
th = np.linspace(0, np.pi / 2, R)
F = rb.Field.build("vert_heterogeneous")
t = time.perf_counter(); b = rb.Batch(F, 6, rb.DELTA_S, 30228, (-2, 5, -2.5, 1), 1, th, -2.0, -2.0, record_stride=1, rec_rows=3072); t_create = time.perf_counter() - t
t = time.perf_counter(); b.run(); t_run = time.perf_counter() - t
t = time.perf_counter(); d = b.d_ray(); f = b.final(); t_small = time.perf_counter() - t
t = time.perf_counter(); s = b.rows(0, 64); t_rows = time.perf_counter() - t
gb = s.nbytes / 1e9
print(f"create (alloc 176 GB + zero + H2D 25 MB + init) {t_create*1e3:.1f} ms; run {t_run*1e3:.1f} ms; d_ray+final D2H (100 MB) {t_small*1e3:.1f} ms; "
      f"64 rows ({gb:.2f} GB) {t_rows*1e3:.1f} ms = {gb/t_rows:.1f} GB/s -> full s_ray (151 GB) ~{151/ (gb/t_rows):.1f} s")
steps = int(d[2].sum())
print(f"ray-steps/s: kernel only {steps/t_run:.3e}; + final-state read-back {steps/(t_run+t_small):.3e}; + full trajectory to host {steps/(t_run + 151/(gb/t_rows)):.3e}")
