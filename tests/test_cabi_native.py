"""The boundary from the C side: tests/native/cabi_smoke.c is compiled with gcc against include/rtmi.h, linked with librtmi.so
and (GPU) run -- field build, batch, loop, checkpoint / resume, read-back -- against numbers the oracle provides; and the
ctypes stub INTEGRATION.md section 2 shows a maintainer is extracted from the document and executed as it stands."""
import os
import re
import struct
import subprocess
import sys

import numpy as np
import pytest

from conftest import LIMITS, ROOT, golden

LIBDIR = os.path.join(ROOT, "raytracing_amd")


def build_c_caller(tmp_path):
    exe = str(tmp_path / "cabi_smoke")
    subprocess.check_call(["gcc", "-std=c11", "-O1", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "native", "cabi_smoke.c"), "-o", exe,
                           "-L", LIBDIR, "-lrtmi", "-lm", f"-Wl,-rpath,{LIBDIR}", "-Wl,-rpath-link,/opt/rocm/lib"])
    return exe


def write_expectations(path, R=64):
    from oracle import rt_oracle as O
    from raytracing_amd import rt_bench as rb
    lim = LIMITS["vert_heterogeneous"]
    th = np.linspace(0.0, np.pi / 2, R)
    ms = int(np.ceil(80 / rb.DELTA_S) + 1)
    o = O.trazar(O.Field("vert_heterogeneous", lim, rb.DELTA), 6, 1, rb.DELTA_S, ms, lim, -2.0, -2.0, th, record_stride=64)
    with open(path, "wb") as f:
        f.write(struct.pack("<qqd", R, ms, rb.DELTA_S))
        for a in (th, o["d_ray"], o["final"], o["s_ray"][1, 4]):
            f.write(np.ascontiguousarray(a, dtype="<f8").tobytes())


def test_c_caller_compiles_and_links_against_the_header(tmp_path):
    exe = build_c_caller(tmp_path)
    assert os.path.exists(exe)
    import torch
    if torch.cuda.is_available():
        return
    # no GPU here: the library reports it (exit 77 = "rtmi_field_build -> RTMI_ERR_HIP"), it does not compute on the CPU
    write_expectations(str(tmp_path / "expect.bin"), R=8)
    r = subprocess.run([exe, str(tmp_path / "expect.bin")], capture_output=True, text=True, timeout=120)
    assert r.returncode == 77, (r.returncode, r.stdout, r.stderr)


@pytest.mark.gpu
def test_c_caller_runs_the_path_and_agrees_with_the_oracle(tmp_path):
    exe = build_c_caller(tmp_path)
    write_expectations(str(tmp_path / "expect.bin"))
    r = subprocess.run([exe, str(tmp_path / "expect.bin")], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, (r.returncode, r.stdout, r.stderr)
    assert "0 step counts differ" in r.stdout and "method 12 -> -1" in r.stdout
    # a plain C process has no torch in it: librtmi.so runs on the HIP runtime its RUNPATH names (/opt/rocm)


def integration_stub_source():
    doc = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    sec = doc[doc.index("## 2."):doc.index("## 3.")]
    m = re.search(r"```python\n(.*?)```", sec, flags=re.S)
    assert m, "INTEGRATION.md section 2 has no python block"
    return m.group(1)


def test_integration_stub_is_valid_python_and_binds_the_current_struct():
    src = integration_stub_source()
    compile(src, "INTEGRATION.md#2", "exec")
    from raytracing_amd import _lib
    fields = re.findall(r'\("(\w+)", C\.c_', src[src.index("class _Params"):src.index("def _ok")])
    assert fields == [n for n, _ in _lib.Params._fields_]          # the document's rtmi_params is the header's, field for field


@pytest.mark.gpu
def test_integration_stub_executes_against_the_reference_fixture():
    """The text a maintainer would paste into RT_bench.py, executed as it stands in a namespace that holds
    what RT_bench.py's module scope holds at that point (constants(), DELTA, N, gamma, op1..op11): trazar_gpu(op6, ...) on the
    vert_heterogeneous preset against the reference's own fixture traj_vert_op6.npz."""
    from raytracing_amd import rt_bench as rb
    src = integration_stub_source().replace('C.CDLL("librtmi.so")', f'C.CDLL({os.path.join(LIBDIR, "librtmi.so")!r})')
    assert "librtmi.so" in src
    ns = {"constants": rb.constants, "DELTA": rb.DELTA, "N": rb.N, "gamma": 1}
    ns.update({f"op{i}": getattr(rb, f"op{i}") for i in range(1, 12)})
    rb._lib.lib()                                   # one HIP runtime in this process (raytracing_amd/_lib.py), then the stub's own CDLL
    exec(compile(src, "INTEGRATION.md#2", "exec"), ns)
    t = golden("traj_vert_op6")
    s_ray, d_ray, ctimes, errors = ns["trazar_gpu"](ns["op6"], None, None, False, float(t["step"]), 91, "3")
    assert s_ray.shape == (int(t["max_size"]), 6, 31) and d_ray.shape == (3, 31)
    assert np.array_equal(d_ray[2], t["d_ray"][2])
    from conftest import sub_rows
    from bench import parity_relerr
    strided, last = sub_rows(s_ray, d_ray, int(t["stride"]))
    assert parity_relerr(strided, t["strided"]) < 1e-9 and parity_relerr(last, t["last"]) < 1e-9
    assert parity_relerr(d_ray[:2], t["d_ray"][:2]) < 1e-9
