#!/usr/bin/env python3
"""Write-only and copy HBM rates of this GPU (torch fill_/copy_ on a buffer the size of the full trajectory record):
the ceiling the row stores of the recording kernel are priced against in DESIGN.md."""
import torch, sys
gb = float(sys.argv[1]) if len(sys.argv) > 1 else 93.0
n = int(gb * 1e9 / 8)
x = torch.empty(n, dtype=torch.float64, device="cuda")
def timed(f, reps=5):
    f(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): f()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps
t = timed(lambda: x.fill_(1.0))
print(f"fill_  {gb:.0f} GB: {t:.2f} ms  {n * 8 / t / 1e9:.2f} TB/s written")
t = timed(lambda: x.zero_())
print(f"zero_  {gb:.0f} GB: {t:.2f} ms  {n * 8 / t / 1e9:.2f} TB/s written")
h = n // 2
t = timed(lambda: x[:h].copy_(x[h:2 * h]))
print(f"copy_  {h * 8 / 1e9:.0f} GB: {t:.2f} ms  {2 * h * 8 / t / 1e9:.2f} TB/s read+written")
