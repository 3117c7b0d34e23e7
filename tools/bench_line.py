#!/usr/bin/env python3
"""Run bench.py with the given arguments and print one compact line (value, ms per pass, kernel ms, VGPRs)."""
import json
import subprocess
import sys

out = subprocess.run([sys.executable, "bench.py", "--cpu-seconds", "0"] + sys.argv[1:], capture_output=True, text=True)
try:
    d = json.loads(out.stdout.strip().splitlines()[-1])
    r = d["roofline"]
    print(f"{' '.join(sys.argv[1:]):70s} {d['value']:.4e} ray-steps/s  {d['ms_per_step']:9.3f} ms/pass  kernel {r['kernel_ms']:9.3f} ms  "
          f"vgpr {r['vgprs']}  steps/pass {d['config']['ray_steps_per_pass_rank0']}")
except Exception as e:   # noqa
    print("bench failed:", e, out.stdout[-2000:], out.stderr[-2000:])
