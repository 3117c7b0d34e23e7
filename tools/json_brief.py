#!/usr/bin/env python3
"""One compact line per bench.py JSON file given: value, ms per pass, kernel ms, schedule, re-traced rays, parity."""
import json
import sys

for path in sys.argv[1:]:
    try:
        d = json.loads(open(path).read().strip().splitlines()[-1])
        c, r, p = d["config"], d["roofline"], d.get("parity_check", {})
        print(f"{path:48s} {d['value']:.4e} ray-steps/s {d['ms_per_step']:8.3f} ms/pass kernel {r['kernel_ms_per_pass']:8.3f} ms {c['launch_mode_used']:7s} "
              f"vgpr {r['vgprs']:3d} frac {r['frac']:.3f} retraced {c.get('retraced')} parity {p.get('max_rel_err')} rows {p.get('rows_max_rel_err')} ok {p.get('ok')} "
              f"explore {c.get('auto_exploration')}")
    except Exception as e:   # noqa
        print(path, "unreadable:", e)
