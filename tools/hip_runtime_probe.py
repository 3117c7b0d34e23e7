#!/usr/bin/env python3
"""Which HIP / HSA runtimes does a process end up with when librtmi.so and torch share it?  (GPU box.)

  python3 tools/hip_runtime_probe.py rtmi_first     librtmi.so (RUNPATH /opt/rocm/lib) loaded and used before torch is imported
  python3 tools/hip_runtime_probe.py torch_first    torch imported and initialised first
  python3 tools/hip_runtime_probe.py preload        torch's libamdhip64.so dlopen'ed (RTLD_GLOBAL) before librtmi.so, torch
                                                    imported afterwards -- what raytracing_amd/_lib.py does

Both copies of the runtime carry SONAME libamdhip64.so.7; torch's libraries ask for "libamdhip64.so" (no version), so the
loader reuses an already-mapped copy only when it is the SAME FILE.  Prints the mapped runtime files after each stage and
whether torch sees the device."""
import ctypes
import importlib.util
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def mapped():
    out = set()
    for line in open("/proc/self/maps"):
        p = line.split()[-1]
        b = os.path.basename(p)
        if b.startswith(("libamdhip64", "libhsa-runtime64", "librccl", "librtmi")):
            out.add(p)
    return sorted(out)


def trace():
    import numpy as np
    from raytracing_amd import rt_bench as rb
    fld = rb.Field.build("vert_heterogeneous")
    b = rb.Batch(fld, rb.op6, rb.DELTA_S, 4000, (-2, 5, -2.5, 1), 1, np.linspace(0, 1.5, 256), -2.0, -2.0, record_stride=1,
                 rec_rows=3072)
    b.run()
    return fld, b


def main():
    mode = sys.argv[1]
    if mode == "rtmi_first":
        os.environ["RTMI_NO_PRELOAD"] = "1"
        fld, b = trace()
        print("after librtmi:", mapped())
        import torch
        print("torch sees a device:", torch.cuda.is_available())
        print("after torch:", mapped())
    elif mode == "torch_first":
        import torch
        torch.cuda.init()
        print("after torch:", mapped())
        fld, b = trace()
        print("after librtmi:", mapped())
    else:
        spec = importlib.util.find_spec("torch")
        p = os.path.join(os.path.dirname(spec.origin), "lib", "libamdhip64.so")
        ctypes.CDLL(p, mode=ctypes.RTLD_GLOBAL)
        print("after preload:", mapped())
        os.environ["RTMI_NO_PRELOAD"] = "1"
        fld, b = trace()
        print("after librtmi:", mapped())
        import torch
        print("torch sees a device:", torch.cuda.is_available())
        print("after torch:", mapped())
    try:
        t = b.device_tensors()
        print("device_tensors ok: s_ray", tuple(t["s_ray"].shape), "istep sum", int(t["istep"].sum().item()), "== stats",
              b.stats()["ray_steps"])
    except Exception as e:
        print("device_tensors FAILED:", e)


if __name__ == "__main__":
    main()
