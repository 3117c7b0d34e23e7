#!/usr/bin/env python3
"""The re-trace kernels ALONE on the chip: the 1 M-ray interface fan stepped with rtmi_step (asynchronous: nobody watches the queue),
then one read -- the queued rays are re-traced in order on the batch's stream after the main kernel has finished.  Under
rocprofv3 --kernel-trace this gives k_retrace_ref's and k_retrace_tail's durations without the main kernel's waves beside them
(compare profiles/r05_iface_retrace_timelines.txt, where they run beside it)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from raytracing_amd import rt_bench as rb          # noqa: E402

m = int(sys.argv[1]) if len(sys.argv) > 1 else 6
lim = (-2, 20, -2, 4)
R = 1 << 20
th = np.linspace(2 * np.pi / 60, np.pi / 2, R)
ms = int(np.ceil(80 / rb.DELTA_S) + 1)
F = rb.Field.build("interface", lim, rb.DELTA)
b = rb.Batch(F, m, rb.DELTA_S, ms, lim, 1, th, -2.0, -2.0, record_stride=0, launch_mode="plain")
for _ in range(2):
    b.reset()
    b.step(ms)
    st = b.stats()
    print(f"op{m}: kernel ms of the pass {st['kernel_ms']:.3f} (main kernel + the re-trace after it), {st['retraced']} rays re-traced", flush=True)
b.close(); F.close()
