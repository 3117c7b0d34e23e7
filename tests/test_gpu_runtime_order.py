"""GPU: raytracing_amd and torch in one process, in either import order, share ONE HIP runtime (raytracing_amd/_lib.py maps
torch's bundled libamdhip64 before librtmi.so; tools/hip_runtime_probe.py shows what happens without that)."""
import os
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu

_PROG = r'''
import os, sys
sys.path.insert(0, {root!r})
import numpy as np
order = sys.argv[1]
if order == "torch_first":
    import torch
    assert torch.cuda.is_available()
from raytracing_amd import rt_bench as rb
assert ("torch" in sys.modules) == (order == "torch_first")
fld = rb.Field.build("vert_heterogeneous")
b = rb.Batch(fld, rb.op6, rb.DELTA_S, 4000, (-2, 5, -2.5, 1), 1, np.linspace(0, 1.5, 256), -2.0, -2.0, record_stride=1, rec_rows=3072)
b.run()                                              # the device has been opened and used through librtmi.so ...
steps = b.stats()["ray_steps"]
import torch                                         # ... before torch is imported (order "rtmi_first")
assert torch.cuda.is_available(), "torch sees no HIP device"
t = b.device_tensors()
assert int(t["istep"].sum().item()) == steps and tuple(t["s_ray"].shape) == (3072, 6, 256)
x = torch.ones(8, device="cuda") * 2                 # torch computes on the same runtime
assert float(x.sum().item()) == 16.0
hip = sorted({{l.split()[-1] for l in open("/proc/self/maps") if os.path.basename(l.split()[-1]).startswith("libamdhip64")}})
hsa = sorted({{l.split()[-1] for l in open("/proc/self/maps") if os.path.basename(l.split()[-1]).startswith("libhsa-runtime64")}})
assert len(hip) == 1 and len(hsa) == 1, (hip, hsa)
print("OK", order, hip[0])
'''


@pytest.mark.parametrize("order", ["rtmi_first", "torch_first"])
@pytest.mark.timeout(600)
def test_one_hip_runtime_in_either_import_order(order, tmp_path):
    prog = tmp_path / "prog.py"
    prog.write_text(_PROG.format(root=ROOT))
    env = {k: v for k, v in os.environ.items() if k != "RTMI_NO_PRELOAD"}
    r = subprocess.run([sys.executable, str(prog), order], capture_output=True, text=True, env=env, timeout=550)
    assert r.returncode == 0 and f"OK {order}" in r.stdout, (r.stdout[-1500:], r.stderr[-3000:])
