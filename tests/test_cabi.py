"""CPU: the C-ABI library loads and exports every symbol include/rtmi.h declares; argument errors are
reported through the status code + rtmi_last_error (no compute without a GPU)."""
import ctypes as C
import os
import re

import pytest

from conftest import ROOT
from raytracing_amd import _lib


def declared_symbols():
    src = open(os.path.join(ROOT, "include", "rtmi.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(rtmi_[a-z0-9_]+)\s*\(", src)))


def test_header_and_binding_agree():
    assert declared_symbols() == sorted(_lib.SYMBOLS)


def test_library_exports_every_declared_symbol():
    L = C.CDLL(_lib.LIB_PATH)
    for name in declared_symbols():
        assert hasattr(L, name), name
    assert _lib.lib().rtmi_abi_version() == _lib.ABI_VERSION == 6


def test_missing_library_fails_loudly(monkeypatch):
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", "/nonexistent/librtmi.so")
    with pytest.raises(ImportError, match="no CPU fallback"):
        _lib.lib()


def test_argument_errors_do_not_touch_the_gpu():
    L = _lib.lib()
    h = C.c_void_p()
    assert L.rtmi_field_build(7, -2.0, 5.0, -2.5, 1.0, 0.0176, 0, None, C.byref(h)) == -1
    assert b"scenario" in L.rtmi_last_error()
    assert L.rtmi_field_build(3, -2.0, 5.0, -2.5, 1.0, -1.0, 0, None, C.byref(h)) == -1
    assert L.rtmi_step(None, 1) == -1 and L.rtmi_run(None) == -1
    assert L.rtmi_batch_stats(None, None) == -1
    with pytest.raises(_lib.RtmiError):
        _lib.check(L.rtmi_read_d_ray(None, None))


def test_ctypes_structs_match_the_header(tmp_path):
    """sizeof of the ctypes mirrors == sizeof of the C structs (gcc on include/rtmi.h)."""
    import subprocess
    src = tmp_path / "sz.c"
    src.write_text('#include "rtmi.h"\n#include <stdio.h>\nint main(void){printf("%zu %zu %zu %zu\\n", sizeof(rtmi_params), '
                   'sizeof(rtmi_stats), sizeof(rtmi_device_view), sizeof(rtmi_shard_stats)); return 0;}\n')
    exe = tmp_path / "sz"
    subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), "-o", str(exe), str(src)])
    sizes = [int(v) for v in subprocess.check_output([str(exe)]).split()]
    assert sizes == [C.sizeof(_lib.Params), C.sizeof(_lib.Stats), C.sizeof(_lib.DeviceView), C.sizeof(_lib.ShardStats)]
