"""ctypes binding of librtmi.so (include/rtmi.h).  No fallback: if the HIP library is missing or a
call fails this raises -- the product path never computes on the CPU."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# RTMI_LIB_PATH: load another build of the same ABI (A/B timing of kernel variants in one session)
LIB_PATH = os.environ.get("RTMI_LIB_PATH") or os.path.join(_HERE, "librtmi.so")

ABI_VERSION = 7
AUTO_SAMPLES = 3        # RTMI_AUTO_SAMPLES (include/rtmi.h)
# rtmi_launch_mode (include/rtmi.h)
LAUNCH_AUTO, LAUNCH_REFILL, LAUNCH_SLICED, LAUNCH_PLAIN = 0, 1, 2, 3
LAUNCH_MODES = {"auto": LAUNCH_AUTO, "refill": LAUNCH_REFILL, "sliced": LAUNCH_SLICED, "plain": LAUNCH_PLAIN, "lane": LAUNCH_PLAIN}
LAUNCH_NAMES = {LAUNCH_AUTO: "auto", LAUNCH_REFILL: "refill", LAUNCH_SLICED: "sliced", LAUNCH_PLAIN: "plain"}

_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int32)


class RtmiError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"librtmi error {code}: {msg}")
        self.code = code


class Params(C.Structure):
    _fields_ = [("method", C.c_int32), ("dtype", C.c_int32), ("gamma", C.c_double), ("gamma_step", C.c_double),
                ("step", C.c_double), ("max_size", C.c_int32), ("record_stride", C.c_int32), ("rec_rows", C.c_int64),
                ("box", C.c_double * 4), ("launch_mode", C.c_int32), ("block_size", C.c_int32),
                ("refill_min", C.c_int32), ("exact_basis", C.c_int32),
                ("field_path", C.c_int32), ("sort_rays", C.c_int32),
                ("ext_s_ray", C.c_void_p), ("ext_n_ray", C.c_void_p), ("lazy_clear", C.c_int32), ("no_n_ray", C.c_int32), ("slice_steps", C.c_int32),
                ("reference_order", C.c_int32), ("no_retrace", C.c_int32)]


class DeviceView(C.Structure):
    _fields_ = [("s_ray", C.c_void_p), ("n_ray", C.c_void_p), ("x", C.c_void_p), ("y", C.c_void_p),
                ("theta", C.c_void_p), ("n", C.c_void_p), ("gx", C.c_void_p), ("gy", C.c_void_p),
                ("dist_sim", C.c_void_p), ("dist_real", C.c_void_p), ("T", C.c_void_p), ("istep", C.c_void_p),
                ("perm", C.c_void_p),
                ("R", C.c_int64), ("rec_rows", C.c_int64), ("dtype", C.c_int32), ("record_stride", C.c_int32)]


class Stats(C.Structure):
    _fields_ = [("ray_steps", C.c_uint64), ("live_rays", C.c_uint64), ("kernel_ms", C.c_double),
                ("launches", C.c_uint32), ("vgprs", C.c_uint32), ("sgprs", C.c_uint32), ("lds_bytes", C.c_uint32),
                ("launch_mode_used", C.c_uint32), ("kernel_ms_total", C.c_double), ("launches_total", C.c_uint64),
                ("auto_fallbacks", C.c_uint32), ("auto_kept", C.c_uint32),
                ("auto_ms", (C.c_double * AUTO_SAMPLES) * 2), ("auto_n", C.c_uint32 * 2),
                ("retraced", C.c_uint32), ("retrace_overflow", C.c_uint32), ("retraced_total", C.c_uint64),
                ("dispatch_first", C.c_uint32), ("reserved_", C.c_uint32)]


class ShardStats(C.Structure):
    _fields_ = [("ndev", C.c_int32), ("transport", C.c_int32), ("R", C.c_int64), ("ray_steps", C.c_uint64), ("live_rays", C.c_uint64),
                ("run_seconds", C.c_double), ("kernel_ms_max", C.c_double), ("auto_fallbacks", C.c_uint32), ("reserved_", C.c_uint32)]


SHARD_AUTO, SHARD_RCCL, SHARD_COPY = 0, 1, 2
# every symbol include/rtmi.h declares: name -> (restype, argtypes)
SYMBOLS = {
    "rtmi_abi_version": (C.c_int, []),
    "rtmi_last_error": (C.c_char_p, []),
    "rtmi_set_device": (C.c_int, [C.c_int]),
    "rtmi_device_count": (C.c_int, [C.POINTER(C.c_int)]),
    "rtmi_field_build": (C.c_int, [C.c_int] + [C.c_double] * 5 + [C.c_int, C.c_void_p, C.POINTER(C.c_void_p)]),
    "rtmi_field_from_samples": (C.c_int, [_dp, C.c_int, _dp, C.c_int, _dp, C.c_double, C.c_int, C.c_void_p,
                                          C.POINTER(C.c_void_p)]),
    "rtmi_field_dims": (C.c_int, [C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "rtmi_field_read": (C.c_int, [C.c_void_p] + [_dp] * 5),
    "rtmi_field_eval": (C.c_int, [C.c_void_p, C.c_int64] + [_dp] * 5),
    "rtmi_field_destroy": (None, [C.c_void_p]),
    "rtmi_batch_create": (C.c_int, [C.c_void_p, C.POINTER(Params), C.c_int64, _dp, _dp, _dp, C.c_void_p,
                                    C.POINTER(C.c_void_p)]),
    "rtmi_batch_set_state": (C.c_int, [C.c_void_p, _dp, _dp, _ip]),
    "rtmi_batch_get_state": (C.c_int, [C.c_void_p, _dp, _dp, _ip, C.c_void_p]),
    "rtmi_batch_restore_state": (C.c_int, [C.c_void_p, _dp, _dp, _ip, C.c_void_p]),
    "rtmi_batch_set_per_ray": (C.c_int, [C.c_void_p, _dp, _ip]),
    "rtmi_batch_reset": (C.c_int, [C.c_void_p]),
    "rtmi_step": (C.c_int, [C.c_void_p, C.c_int32]),
    "rtmi_step_repeat": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32]),
    "rtmi_run": (C.c_int, [C.c_void_p]),
    "rtmi_sync": (C.c_int, [C.c_void_p]),
    "rtmi_read_d_ray": (C.c_int, [C.c_void_p, _dp]),
    "rtmi_read_final": (C.c_int, [C.c_void_p, _dp]),
    "rtmi_read_rows": (C.c_int, [C.c_void_p, C.c_int64, C.c_int64, _dp, _dp]),
    "rtmi_metric": (C.c_int, [C.c_void_p, C.c_int, _dp]),
    "rtmi_isochrones": (C.c_int, [C.c_void_p, C.c_int32, _dp, _dp]),
    "rtmi_wavefronts": (C.c_int, [C.c_void_p, C.c_int32, _dp, C.c_int32, C.POINTER(C.c_int64), _dp, _dp]),
    "rtmi_batch_view": (C.c_int, [C.c_void_p, C.POINTER(DeviceView)]),
    "rtmi_batch_stats": (C.c_int, [C.c_void_p, C.POINTER(Stats)]),
    "rtmi_batch_destroy": (None, [C.c_void_p]),
    "rtmi_shard_create": (C.c_int, [C.c_int] + [C.c_double] * 5 + [C.POINTER(Params), C.c_int64, _dp, _dp, _dp, _ip, C.c_int, C.c_int,
                                    C.POINTER(C.c_void_p)]),
    "rtmi_shard_run": (C.c_int, [C.c_void_p]),
    "rtmi_shard_reset": (C.c_int, [C.c_void_p]),
    "rtmi_shard_read_d_ray": (C.c_int, [C.c_void_p, _dp]),
    "rtmi_shard_read_final": (C.c_int, [C.c_void_p, _dp]),
    "rtmi_shard_gather_rows": (C.c_int, [C.c_void_p, C.c_int64, C.c_int64, C.c_int64, C.POINTER(C.c_void_p)]),
    "rtmi_shard_read_rows": (C.c_int, [C.c_void_p, C.c_int64, C.c_int64, C.c_int64, _dp]),
    "rtmi_shard_info": (C.c_int, [C.c_void_p, C.POINTER(ShardStats)]),
    "rtmi_shard_batch": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_int)]),
    "rtmi_shard_destroy": (None, [C.c_void_p]),
    "rtmi_debug_sincos": (C.c_int, [C.c_int64, _dp, _dp, _dp]),
    "rtmi_debug_field_lookup": (C.c_int, [C.c_void_p, C.c_int64] + [_dp] * 5),
    "rtmi_debug_auto_rule": (C.c_int, [_dp, C.c_int, _dp, C.c_int, _ip, _ip]),
}

_lib = None


def _share_torchs_hip_runtime():
    """One HIP runtime per process, whatever the import order.

    librtmi.so asks for "libamdhip64.so.7" (RUNPATH /opt/rocm/lib); torch's ROCm wheel bundles a copy of the runtime with the
    same SONAME, and its libraries ask for "libamdhip64.so" -- a name the loader satisfies from an already mapped object only
    when that object is the same FILE.  So: torch first, librtmi second -> librtmi's request matches the SONAME of torch's copy,
    one runtime (fine); librtmi first -> /opt/rocm's copy is mapped, torch later maps its own beside it, and the second HSA
    runtime to open the device is refused ("No HIP GPUs are available"; tools/hip_runtime_probe.py shows both cases).
    Mapping torch's copy (and nothing else of torch) before librtmi.so makes every order the first one: librtmi binds to it by
    SONAME, torch finds the same file when it is imported.  Without torch installed librtmi.so uses /opt/rocm's runtime.
    RTMI_NO_PRELOAD=1 skips this (the probe uses it)."""
    import importlib.util
    import sys
    if os.environ.get("RTMI_NO_PRELOAD"):
        return
    if "torch" in sys.modules:
        # torch imported first: its runtime is mapped once torch has initialised its HIP side; do that before librtmi.so asks
        # for "libamdhip64.so.7", so that the request is answered by SONAME from torch's copy and not from /opt/rocm
        try:
            sys.modules["torch"].cuda.init()
        except Exception:
            pass                                   # no device / a CPU-only torch: librtmi's own calls will say so
        return
    try:
        spec = importlib.util.find_spec("torch")                           # locates the package, does not import it
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.origin:
        return
    hip = os.path.join(os.path.dirname(spec.origin), "lib", "libamdhip64.so")
    if os.path.exists(hip):
        C.CDLL(hip, mode=C.RTLD_GLOBAL)


def mapped_hip_runtimes():
    """Paths of every libamdhip64 mapped into this process (/proc/self/maps)."""
    found = []
    try:
        with open("/proc/self/maps") as fh:
            for line in fh:
                path = line.rsplit(None, 1)[-1] if "/" in line else ""
                if "libamdhip64" in os.path.basename(path) and path not in found:
                    found.append(path)
    except OSError:
        pass
    return found


def _check_one_hip_runtime():
    """Two HIP runtimes in one process (a torch wheel built for another ROCm major version, so that the SONAMEs differ, or
    RTMI_NO_PRELOAD set) fail much later and far from the cause -- "No HIP GPUs are available" from whichever runtime opens the
    device second.  Say so here, by name, at load time."""
    import warnings
    paths = mapped_hip_runtimes()
    if len({os.path.realpath(p) for p in paths}) > 1:
        msg = ("raytracing_amd: more than one HIP runtime is mapped into this process (" + ", ".join(paths) + "): librtmi.so and "
               "torch would each open the device through their own; the second to do so gets 'No HIP GPUs are available'. "
               "Import raytracing_amd before torch without RTMI_NO_PRELOAD, or use a torch wheel of librtmi's ROCm major version.")
        if os.environ.get("RTMI_STRICT_RUNTIME"):
            raise ImportError(msg)
        warnings.warn(msg, RuntimeWarning, stacklevel=3)


def lib():
    """Load librtmi.so (built in-tree by __graft_entry__.build() / make -C raytracing_amd/csrc)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                f"{LIB_PATH} is missing: build the HIP extension first "
                "(python -c 'import __graft_entry__ as g; g.build()' or make -C raytracing_amd/csrc). "
                "raytracing_amd has no CPU fallback.")
        _share_torchs_hip_runtime()
        L = C.CDLL(LIB_PATH)
        _check_one_hip_runtime()
        for name, (res, args) in SYMBOLS.items():
            fn = getattr(L, name)  # AttributeError if the library does not export a declared symbol
            fn.restype = res
            fn.argtypes = args
        if L.rtmi_abi_version() != ABI_VERSION:
            raise ImportError("librtmi.so ABI version mismatch")
        _lib = L
    return _lib


def check(code):
    if code != 0:
        raise RtmiError(code, lib().rtmi_last_error().decode("utf-8", "replace"))


def dptr(a):
    return a.ctypes.data_as(_dp) if a is not None else None
