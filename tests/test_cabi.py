"""CPU: the C-ABI library loads and exports every symbol include/rtmi.h declares; argument errors are
reported through the status code + rtmi_last_error (no compute without a GPU)."""
import ctypes as C
import numpy as np
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
    assert _lib.lib().rtmi_abi_version() == _lib.ABI_VERSION == 7


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


def _auto_rule(sliced, plain):
    L = _lib.lib()
    nxt, dec = C.c_int32(-7), C.c_int32(-7)
    a = np.asarray(sliced, dtype=np.float64); b = np.asarray(plain, dtype=np.float64)
    _lib.check(L.rtmi_debug_auto_rule(_lib.dptr(a) if len(a) else None, len(a), _lib.dptr(b) if len(b) else None, len(b),
                                      C.byref(nxt), C.byref(dec)))
    return nxt.value, dec.value


def test_launch_auto_rule_on_recorded_sequences():
    """RTMI_LAUNCH_AUTO's decision rule (host code, no GPU): interleaved and mirrored exploration order, medians, the plain
    launch kept only when more than 3 % ahead -- on sequences like the ones boxes produced (kernel ms per pass)."""
    S, P = _lib.LAUNCH_SLICED, _lib.LAUNCH_PLAIN
    # exploration order: sliced, plain, plain, sliced, sliced, plain, then over
    order, ns, np_ = [], 0, 0
    while True:
        nxt, _ = _auto_rule([15.0] * ns, [15.0] * np_)
        if nxt < 0:
            break
        order.append(nxt)
        ns, np_ = ns + (nxt == S), np_ + (nxt == P)
    assert order == [S, P, P, S, S, P] and (ns, np_) == (_lib.AUTO_SAMPLES, _lib.AUTO_SAMPLES)
    # the headline on a box whose clock is still falling through the exploration (each pass slower than the one before, the
    # first one cold): round 4's rule compared one early plain sample with a later sliced one and kept the plain launch, which
    # then ran 16.3 ms where the sliced one runs 15.3.  In order of execution: S 16.4 (cold), P 15.35, P 15.6, S 15.45, S 15.7, P 16.1
    assert _auto_rule([16.4, 15.45, 15.7], [15.35, 15.6, 16.1])[1] == S
    # the headline on a steady box: slicing 2 % ahead
    assert _auto_rule([15.9, 15.3, 15.35], [15.7, 15.65, 15.6])[1] == S
    # the interface fan: the plain launch 5 % ahead on every sample -> plain
    assert _auto_rule([23.4, 23.3, 23.35], [22.2, 22.15, 22.3])[1] == P
    # ahead by less than the hysteresis (2 %): slicing stays
    assert _auto_rule([10.0, 10.0, 10.0], [9.8, 9.8, 9.8])[1] == S
    # one wild sample on either side does not decide (medians)
    assert _auto_rule([15.3, 40.0, 15.3], [15.6, 15.6, 9.0])[1] == S
    assert _auto_rule([23.3, 23.3, 2.0], [22.0, 80.0, 22.0])[1] == P
    # cfg3 (fisheye): slicing 25 % ahead
    assert _auto_rule([8.3, 8.0, 8.1], [10.9, 10.8, 10.8])[1] == S
    # bad arguments do not crash
    assert _lib.lib().rtmi_debug_auto_rule(None, 5, None, 0, None, None) == -1
