"""Host API for the MI355X ray propagation path -- the selection surface of neyuru/RayTracing's
RT_bench.py (scenarios, step methods op1..op11, DELTA_S stepping, trazar) over librtmi.so.

Names, argument order and return shapes follow the reference so a caller of
`genZ` / `interpolacion` / `n_gradient` / `trazar` / `search_delta` can switch imports; all numerics
run in HIP kernels on the GPU (see include/rtmi.h).  Reference lines are RT_bench.py file:line.
There is no CPU path here: without a HIP device every compute call raises.
"""
import ctypes as C
import time

import numpy as np

from . import _lib
from ._lib import Params, Stats, DeviceView, check, dptr, lib, LAUNCH_MODES, LAUNCH_NAMES, LAUNCH_AUTO, LAUNCH_REFILL, LAUNCH_SLICED, LAUNCH_PLAIN

# re-runs of one batch (reset + run) RTMI_LAUNCH_AUTO spends timing its two schedules before it keeps one (rtmi.h: RTMI_AUTO_SAMPLES each)
AUTO_EXPLORE_RUNS = 2 * _lib.AUTO_SAMPLES

# --------------------------------------------------------------------------- constants (:59-97)
THCK_PARAM = 0.005                                   # :59
SIGMA = 0.05293304824724534                          # :60-61 (value of -2*THCK*log((A-1)/(sqrt2-A)) under numpy)
DELTA_G = np.pi / 2                                  # :64
GOLD_RATIO = (np.sqrt(5) - 1) / 2                    # :65
GOLD_TOL = 1.4901161193847656e-08                    # :66 sqrt(eps)
MAX_DEVIATION = 0.2                                  # :69
DELTA = SIGMA / 3                                    # :77
DELTA_S_DIVISOR = 20                                 # :79
DELTA_S = SIGMA / DELTA_S_DIVISOR                    # :81
N = 10                                               # :82
DELTA_S_DIVISOR_FISHEYE = 90                         # :84
DELTA_STEP = 0.01                                    # :89
DELTA_S_DIVISOR_UPPER_LIMIT = 3                      # :90
DELTA_S_DIVISOR_LOWER_LIMIT = 1 + DELTA_STEP         # :91
DELTA_STEP_FISHEYE = 1                               # :92
DELTA_S_DIVISOR_FISHEYE_UPPER_LIMIT = 303            # :93
DELTA_S_DIVISOR_FISHEYE_LOWER_LIMIT = 4              # :94
DELTA_STEP_VERT = 0.005                              # :95
DELTA_S_DIVISOR_VERT_UPPER_LIMIT = 2                 # :96
DELTA_S_DIVISOR_VERT_LOWER_LIMIT = 1 / 40            # :97

F64, F32 = 0, 1
ORDERS = {"default": 0, "reference": 1, "fused": 2, "fast_field": 3}          # rtmi_order (rtmi_params.reference_order)


# --------------------------------------------------------------------------- scenarios (:106-119)
class Scenario:
    """A scenario token.  genZ() samples it on the device; calling it evaluates the same formula with
    numpy for callers that only want to look at n(x, y) (plots, docs) -- it is not on the trace path."""

    def __init__(self, name, code, fn):
        self.__name__ = name
        self.code = code
        self._fn = fn

    def __call__(self, a, b):
        return self._fn(np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64))

    def __repr__(self):
        return f"<scenario {self.__name__}>"


interface = Scenario("interface", 1, lambda a, b: np.sqrt(2) - (np.sqrt(2) - 1) / (1 + np.exp(-b / THCK_PARAM)))
fisheye = Scenario("fisheye", 2, lambda a, b: 1 / (1 + np.power(a, 2) + np.power(b, 2)))
vert_heterogeneous = Scenario("vert_heterogeneous", 3, lambda a, b: 1 / (18 + 2 * b))
SCENARIOS = {"interface": interface, "fisheye": fisheye, "vert_heterogeneous": vert_heterogeneous,
             "anisotropy": vert_heterogeneous}          # scenario 4 reuses the field of 3 (:1579)
USER_CHOICE = {"interface": "1", "fisheye": "2", "vert_heterogeneous": "3", "anisotropy": "4"}

# module globals the reference's __main__ sets (:1567-1584); genZ reads `f` like the reference does (:432)
f = None
gamma = 1


def anisotropy(theta, gamma):  # :118-119 (host helper; the device evaluates its own copy)
    return np.sqrt((gamma * np.sin(theta)) ** 2 + np.cos(theta) ** 2)


def constants(user_choice):
    """Scenario presets, same 13-tuple as RT_bench.py:247-295."""
    if user_choice == "1":
        g, ray_count = 1, 42
        theta_v = np.linspace(2 * (np.pi / 60), np.pi / 2, ray_count + 1)   # Q9: one unused sample
        pos_x = np.ones(ray_count) * -2
        s = 80
        lim = (-2, 20, -2, 4)
        flags = (1, 0, 0, 0)
    elif user_choice == "2":
        g, ray_count = 1, 1
        theta_v = np.linspace(np.pi / 2, np.pi / 2, 1)
        pos_x = np.array((1, 0))
        s = N * (2 * np.pi)
        lim = (-1.5, 1.5, -1.5, 1.5)
        flags = (0, 1, 0, 0)
    elif user_choice in ("3", "4"):
        g = 1 if user_choice == "3" else 3
        ray_count = 31
        theta_v = np.linspace(0, np.pi / 2, ray_count)
        pos_x = np.ones(ray_count) * -2
        s = 80
        lim = (-2, 5, -2.5, 1)
        flags = (0, 0, 1, 0) if user_choice == "3" else (0, 0, 0, 1)
    else:
        raise ValueError("user_choice must be '1', '2', '3' or '4'")
    return (g, ray_count, theta_v, pos_x, s) + lim + flags


# --------------------------------------------------------------------------- step-method tokens (:469-764)
class StepMethod:
    """Token for op<m>.  Passing it to trazar selects the device kernel; calling it advances one ray by
    one DELTA_S step on the device with the reference's argument list (:469)."""

    def __init__(self, m, label):
        self.method = m
        self.__name__ = f"op{m}"
        self.label = label

    def __repr__(self):
        return f"<step method op{self.method}:{self.label}>"

    def __call__(self, i_angle, init_n, i_grad, i_unitv, i_vpos, coef_i, grd, z, step, history=None):
        fld = _field_of(z, grd)
        g = gamma
        st = np.zeros((9, 1))
        st[:, 0] = (i_vpos[0], i_vpos[1], i_angle, init_n, i_grad[0], i_grad[1], 0.0, 0.0, 0.0)
        hist = None
        if self.method == 7:
            if history is None:
                raise ValueError("op7 needs history=[P0, P1] (the two positions before i_vpos; VECTOR_LIST, :73)")
            hist = np.ascontiguousarray(np.asarray(history, dtype=np.float64).reshape(4, 1))
        b = Batch(fld, self.method, step, max_size=1 << 20, box=(-1e300, 1e300, -1e300, 1e300), gamma=g,
                  thetas=[i_angle], x0=[i_vpos[0]], y0=[i_vpos[1]], record_stride=0)
        b.set_state(st, hist, np.array([3], dtype=np.int32))
        b.step(1)
        fin = b.final()[:, 0]
        b.close()
        return np.array((fin[0], fin[1])), fin[2], fin[3], np.array((fin[4], fin[5]))


_LABELS = ["1st order Taylor + analytical 2-point momentum-impulse", "1st order Taylor + d_theta/d_s Runge-Kutta (AnDF)",
           "2-point curvature + d_theta/d_s Runge-Kutta", "2-point curvature + analytical 2-point momentum-impulse",
           "2-point curvature + optimized 2-point momentum-impulse", "2nd order Taylor + d_theta/d_s Runge-Kutta (HySA)",
           "2nd order Taylor + 4-point difference method (MxSA)", "2nd order Taylor + analytical 2-point momentum-impulse",
           "2nd order Taylor + optimized 2-point momentum-impulse",
           "2-point curvature + optimized anisotropic momentum-impulse",
           "2nd order Taylor + optimized anisotropic momentum-impulse"]
op1, op2, op3, op4, op5, op6, op7, op8, op9, op10, op11 = [StepMethod(i + 1, _LABELS[i]) for i in range(11)]
METHODS = {m.method: m for m in (op1, op2, op3, op4, op5, op6, op7, op8, op9, op10, op11)}
# menus: isotropic scenarios offer 1..9 (:1238-1264), the anisotropic one offers 1..2 -> op10/op11 (:1286-1291)
ISOTROPIC_MENU = {str(i): METHODS[i] for i in range(1, 10)}
ANISOTROPIC_MENU = {"1": op10, "2": op11}


def _method_id(selected_func):
    if isinstance(selected_func, StepMethod):
        return selected_func.method
    if isinstance(selected_func, int) and 1 <= selected_func <= 11:
        return selected_func
    name = getattr(selected_func, "__name__", "")
    if name.startswith("op") and name[2:].isdigit() and 1 <= int(name[2:]) <= 11:
        return int(name[2:])
    raise ValueError(f"selected_func must be one of op1..op11, got {selected_func!r}")


# --------------------------------------------------------------------------- field (:412-464)
class Field:
    """z + grd of interpolacion(), resident in HBM (rtmi_field)."""

    def __init__(self, handle, dtype):
        self._h = C.c_void_p(handle)
        self.dtype = dtype
        qx, qy = C.c_int(), C.c_int()
        check(lib().rtmi_field_dims(self._h, C.byref(qx), C.byref(qy)))
        self.qx, self.qy = qx.value, qy.value

    @classmethod
    def build(cls, scenario, limits=None, delta=DELTA, dtype=F64, stream=None):
        """genZ + interpolacion on the device for one of the four scenarios (rtmi_field_build)."""
        sc = SCENARIOS[scenario] if isinstance(scenario, str) else scenario
        if limits is None:
            limits = constants(USER_CHOICE[scenario])[5:9]
        h = C.c_void_p()
        check(lib().rtmi_field_build(sc.code, *[float(v) for v in limits], float(delta), dtype, stream, C.byref(h)))
        return cls(h.value, dtype)

    @classmethod
    def from_samples(cls, x, y, Z, delta=DELTA, dtype=F64, stream=None):
        x = np.ascontiguousarray(x, dtype=np.float64)
        y = np.ascontiguousarray(y, dtype=np.float64)
        Z = np.ascontiguousarray(Z, dtype=np.float64)
        if Z.shape != (len(y), len(x)):
            raise ValueError("Z must have shape (len(y), len(x))")
        h = C.c_void_p()
        check(lib().rtmi_field_from_samples(dptr(x), len(x), dptr(y), len(y), dptr(Z), float(delta), dtype, stream,
                                            C.byref(h)))
        return cls(h.value, dtype)

    def arrays(self):
        """(x, y, Z, coef_dy, coef_dx): axes, n samples and the bicubic coefficients of GradX/GradY."""
        x = np.empty(self.qx); y = np.empty(self.qy)
        Z = np.empty((self.qy, self.qx)); cdy = np.empty_like(Z); cdx = np.empty_like(Z)
        check(lib().rtmi_field_read(self._h, dptr(x), dptr(y), dptr(Z), dptr(cdy), dptr(cdx)))
        return x, y, Z, cdy, cdx

    def n_gradient(self, x, y):
        x = np.ascontiguousarray(np.atleast_1d(x), dtype=np.float64)
        y = np.ascontiguousarray(np.atleast_1d(y), dtype=np.float64)
        n = np.empty_like(x); gx = np.empty_like(x); gy = np.empty_like(x)
        check(lib().rtmi_field_eval(self._h, len(x), dptr(x), dptr(y), dptr(n), dptr(gx), dptr(gy)))
        return n, gx, gy

    def lookup_fast(self, x, y):
        """The fast-form step methods' own lookup (the cell's polynomial, rtmi_debug_field_lookup) -> n, dn/dx, dn/dy."""
        x = np.ascontiguousarray(np.atleast_1d(x), dtype=np.float64)
        y = np.ascontiguousarray(np.atleast_1d(y), dtype=np.float64)
        n = np.empty_like(x); gx = np.empty_like(x); gy = np.empty_like(x)
        check(lib().rtmi_debug_field_lookup(self._h, len(x), dptr(x), dptr(y), dptr(n), dptr(gx), dptr(gy)))
        return n, gx, gy

    def close(self):
        if self._h:
            lib().rtmi_field_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class FieldSpline:
    """One of the three callables interpolacion() returns (z, grd[0], grd[1]); `spl(y, x)` evaluates on
    the device with RectBivariateSpline's argument order and [[value]] return shape (:153-155)."""

    def __init__(self, field, which):
        self.field, self.which = field, which

    def __call__(self, yv, xv):
        n, gx, gy = self.field.n_gradient(np.atleast_1d(xv), np.atleast_1d(yv))
        return {"n": n, "dx": gx, "dy": gy}[self.which].reshape(-1, 1) if np.ndim(xv) else \
            np.array([[{"n": n, "dx": gx, "dy": gy}[self.which][0]]])


def genZ(xi, xs, yi, ys, dtype=F64):
    """RT_bench.py:412-433.  Reads the module-global scenario `f` like the reference.  Returns
    (x, y, X, Y, ZZ) as numpy arrays; ZZ is sampled on the device for the built-in scenarios."""
    if f is None:
        raise RuntimeError("set raytracing_amd.rt_bench.f to a scenario (interface, fisheye, vert_heterogeneous) first")
    qx = int((xs - xi + 6) / DELTA + 1)
    qy = int((ys - yi + 6) / DELTA + 1)
    x, y = np.linspace(xi - 3, xs + 3, qx), np.linspace(yi - 3, ys + 3, qy)
    X, Y = np.meshgrid(x, y)
    if isinstance(f, Scenario):
        fld = Field.build(f, (xi, xs, yi, ys), DELTA, dtype)
        ZZ = fld.arrays()[2]
        fld.close()
    else:
        ZZ = np.asarray(f(X, Y), dtype=np.float64)   # user-supplied n(x, y): sampled by the caller's function
    return x, y, X, Y, ZZ


def interpolacion(x, y, Z, X=None, Y=None, dtype=F64):
    """RT_bench.py:435-464 -> (z, grd, hess).  The fits run on the device; hess is None (the reference
    builds it and never reads it, :462)."""
    fld = Field.from_samples(x, y, Z, DELTA, dtype)
    return FieldSpline(fld, "n"), (FieldSpline(fld, "dy"), FieldSpline(fld, "dx")), None


def _field_of(z, grd=None):
    if isinstance(z, Field):
        return z
    if isinstance(z, FieldSpline):
        return z.field
    raise TypeError("z must come from interpolacion() / Field.build()")


def n_gradient(vector, grd, z):
    """RT_bench.py:141-156."""
    n, gx, gy = _field_of(z, grd).n_gradient(vector[0], vector[1])
    return n[0], np.array([gx[0], gy[0]])


# --------------------------------------------------------------------------- ray batch (:766-948)
class Batch:
    """One trazar() call's rays, resident in HBM (rtmi_batch)."""

    def __init__(self, field, method, step, max_size, box, gamma, thetas, x0, y0, record_stride=1, rec_rows=0,
                 gamma_step=None, stream=None, ext_s_ray=None, ext_n_ray=None, block_size=0, launch_mode="auto",
                 refill_min=0, exact_basis=0, field_path=0, sort_rays=False, lazy_clear=False, keep_n_ray=True, slice_steps=0,
                 reference_order=False, retrace=True):
        self.field = field
        th = np.ascontiguousarray(thetas, dtype=np.float64)
        self.R = len(th)
        x0 = np.ascontiguousarray(np.broadcast_to(np.asarray(x0, dtype=np.float64), (self.R,)))
        y0 = np.ascontiguousarray(np.broadcast_to(np.asarray(y0, dtype=np.float64), (self.R,)))
        p = Params()
        p.method = _method_id(method); p.dtype = field.dtype
        p.gamma = float(gamma); p.gamma_step = float(gamma if gamma_step is None else gamma_step)
        p.step = float(step); p.max_size = int(max_size)
        p.record_stride = int(record_stride); p.rec_rows = int(rec_rows)
        for i in range(4):
            p.box[i] = float(box[i])
        # launch_mode: "auto" (default: the library chooses and, on re-runs, keeps the faster), "plain", "refill", "sliced",
        # or the rtmi_launch_mode integer; results are bit-identical in all of them
        p.launch_mode = LAUNCH_MODES[launch_mode] if isinstance(launch_mode, str) else int(launch_mode)
        p.block_size = int(block_size); p.refill_min = int(refill_min)
        p.exact_basis = int(exact_basis); p.field_path = int(field_path)
        if isinstance(sort_rays, str):
            if sort_rays != "auto":
                raise ValueError("sort_rays must be True, False or 'auto'")
            sort_rays = not launch_is_coherent(x0, y0, th)
        p.sort_rays = int(bool(sort_rays))
        p.ext_s_ray = ext_s_ray; p.ext_n_ray = ext_n_ray
        p.lazy_clear = int(bool(lazy_clear))
        p.no_n_ray = int(not keep_n_ray)
        p.slice_steps = int(slice_steps)
        # rtmi_order: False / 0 default (op7 alone of the fused five steps in the reference's operation order); True / 1 all of
        # op1/2/6/7/8 (fp64: the oracle's bits, slower); "fused" / 2 fused forms throughout, op7 included; "fast_field" / 3 op7's
        # reference-order step on the fused field lookup
        p.reference_order = ORDERS[reference_order] if isinstance(reference_order, str) else int(reference_order)
        # retrace (default on): a fused fp64 op1/2/6/8 batch re-traces its critical rays -- those running along a sharp
        # transition of the medium -- in reference order by itself (rtmi_params.no_retrace)
        p.no_retrace = int(not retrace)
        self.params = p
        self._h = C.c_void_p()
        check(lib().rtmi_batch_create(field._h, C.byref(p), self.R, dptr(x0), dptr(y0), dptr(th), stream,
                                      C.byref(self._h)))
        self.max_size = int(max_size)
        self.record_stride = int(record_stride)
        self.rec_rows = int(rec_rows) if rec_rows else ((self.max_size + record_stride - 1) // record_stride
                                                        if record_stride else 0)

    def set_state(self, state9, hist4=None, istep=None):
        st = np.ascontiguousarray(state9, dtype=np.float64)
        assert st.shape == (9, self.R)
        h = np.ascontiguousarray(hist4, dtype=np.float64) if hist4 is not None else None
        i = np.ascontiguousarray(istep, dtype=np.int32) if istep is not None else None
        check(lib().rtmi_batch_set_state(self._h, dptr(st), dptr(h), i.ctypes.data_as(_lib._ip) if i is not None else None))

    def get_state(self):
        """(state9 [9,R], aux4 [4,R], istep [R], alive [R]) -- rtmi_batch_get_state; restore_state(*get_state()) on a batch
        with the same parameters resumes bit for bit."""
        st = np.empty((9, self.R)); h = np.empty((4, self.R)); i = np.empty(self.R, dtype=np.int32)
        al = np.empty(self.R, dtype=np.uint8)
        check(lib().rtmi_batch_get_state(self._h, dptr(st), dptr(h), i.ctypes.data_as(_lib._ip), al.ctypes.data))
        return st, h, i, al

    def restore_state(self, state9, aux4, istep, alive):
        st = np.ascontiguousarray(state9, dtype=np.float64)
        h = np.ascontiguousarray(aux4, dtype=np.float64)
        i = np.ascontiguousarray(istep, dtype=np.int32)
        al = np.ascontiguousarray(alive, dtype=np.uint8)
        assert st.shape == (9, self.R) and h.shape == (4, self.R) and i.shape == (self.R,) and al.shape == (self.R,)
        check(lib().rtmi_batch_restore_state(self._h, dptr(st), dptr(h), i.ctypes.data_as(_lib._ip), al.ctypes.data))

    def set_per_ray(self, step, max_size):
        """Per-ray DELTA_S and max_size ([R] each, caller's ray order): rtmi_batch_set_per_ray."""
        st = np.ascontiguousarray(np.broadcast_to(np.asarray(step, dtype=np.float64), (self.R,)))
        ms = np.ascontiguousarray(np.broadcast_to(np.asarray(max_size, dtype=np.int32), (self.R,)))
        check(lib().rtmi_batch_set_per_ray(self._h, dptr(st), ms.ctypes.data_as(_lib._ip)))

    def reset(self):
        check(lib().rtmi_batch_reset(self._h))

    def step(self, nsteps=1, count=1):
        """Advance every live ray by nsteps DELTA_S steps; count > 1: that many such launches as one hipGraph."""
        if count > 1:
            check(lib().rtmi_step_repeat(self._h, int(nsteps), int(count)))
        else:
            check(lib().rtmi_step(self._h, int(nsteps)))

    def run(self):
        check(lib().rtmi_run(self._h))

    def sync(self):
        check(lib().rtmi_sync(self._h))

    def d_ray(self):
        d = np.empty((3, self.R))
        check(lib().rtmi_read_d_ray(self._h, dptr(d)))
        return d

    def final(self):
        out = np.empty((9, self.R))
        check(lib().rtmi_read_final(self._h, dptr(out)))
        return out

    def rows(self, row0=0, nrows=None, want_n_ray=False):
        nrows = self.rec_rows - row0 if nrows is None else nrows
        s = np.empty((nrows, 6, self.R))
        n = np.empty((nrows, self.R)) if want_n_ray else None
        check(lib().rtmi_read_rows(self._h, row0, nrows, dptr(s), dptr(n)))
        return (s, n) if want_n_ray else s

    def metric(self, kind):
        """On-device per-ray metric: "snell" (degrees), "closure" (%), "px_cv" (%); see rtmi_metric."""
        out = np.empty(self.R)
        check(lib().rtmi_metric(self._h, {"snell": 1, "closure": 2, "px_cv": 3}[kind], dptr(out)))
        return out

    def isochrones(self, times):
        """(x, y, theta) of every ray at the given traveltimes -> [ntimes, 3, R], NaN where not reached."""
        t = np.ascontiguousarray(times, dtype=np.float64)
        out = np.empty((len(t), 3, self.R))
        check(lib().rtmi_isochrones(self._h, len(t), dptr(t), dptr(out)))
        return out

    def wavefronts(self, times, nfine=100):
        """The reference's wavefront extraction (RT_bench.py:1005-1044) on the device: one dict per traveltime with the
        points of the wavefront sorted by y -- 'y', 'x', 'angle' (ray angle), 'dxdy' (derivative of the PCHIP interpolant
        x(y) at the points), 'normal' (normal angle), 'angle_diff' (|ray angle - normal angle|), 'ray' (ray indices) --
        and 'x_fine', 'y_fine' (the interpolated wavefront on nfine points).  Wavefronts with < 2 points have empty
        derived arrays, like the reference, which skips them (:1011)."""
        t = np.ascontiguousarray(times, dtype=np.float64)
        nt = len(t)
        count = np.zeros(nt, dtype=np.int64)
        nodes = np.empty((nt, 7, self.R))
        fine = np.empty((nt, 2, nfine)) if nfine else None
        check(lib().rtmi_wavefronts(self._h, nt, dptr(t), int(nfine), count.ctypes.data_as(C.POINTER(C.c_int64)), dptr(nodes),
                                    dptr(fine)))
        out = []
        for i in range(nt):
            n = int(count[i])
            d = dict(time=float(t[i]), count=n, y=nodes[i, 0, :n].copy(), x=nodes[i, 1, :n].copy(), angle=nodes[i, 2, :n].copy(),
                     ray=nodes[i, 6, :n].astype(np.int64))
            m = n if n >= 2 else 0
            d.update(dxdy=nodes[i, 3, :m].copy(), normal=nodes[i, 4, :m].copy(), angle_diff=nodes[i, 5, :m].copy(),
                     x_fine=fine[i, 0].copy() if nfine and m else np.empty(0), y_fine=fine[i, 1].copy() if nfine and m else np.empty(0))
            out.append(d)
        return out

    def stats(self):
        s = Stats()
        check(lib().rtmi_batch_stats(self._h, C.byref(s)))
        out = {k: getattr(s, k) for k, _ in Stats._fields_ if not k.startswith("auto_") and k != "reserved_"}     # incl. retraced, retrace_overflow, dispatch_first
        out["launch_mode_used"] = LAUNCH_NAMES.get(out["launch_mode_used"], out["launch_mode_used"])
        out["auto_fallbacks"] = s.auto_fallbacks
        # RTMI_LAUNCH_AUTO's exploration record: kernel ms of each timed run per schedule, and the schedule kept (None: still exploring / no choice)
        out["auto_exploration"] = {"sliced_ms": [s.auto_ms[0][i] for i in range(s.auto_n[0])],
                                   "plain_ms": [s.auto_ms[1][i] for i in range(s.auto_n[1])],
                                   "kept": LAUNCH_NAMES.get(s.auto_kept) if s.auto_kept else None}
        return out

    def view(self):
        v = DeviceView()
        check(lib().rtmi_batch_view(self._h, C.byref(v)))
        return v

    def device_tensors(self):
        """Zero-copy torch views of the batch's device memory (rtmi_batch_view through
        __cuda_array_interface__): SoA state [R], istep [R], s_ray [rec_rows, 6, R], n_ray [rec_rows, R].
        For consumers that stay on the GPU or feed torch.distributed (RCCL) collectives; the tensors alias
        library memory and die with the batch."""
        import torch
        if not torch.cuda.is_available():
            raise RuntimeError("torch sees no HIP device (raytracing_amd._lib maps torch's own HIP runtime before librtmi.so so "
                               "that the two share one; RTMI_NO_PRELOAD=1 or an RTMI_LIB_PATH build linked elsewhere defeats that)")
        v = self.view()
        ts = "<f8" if v.dtype == F64 else "<f4"

        class _Cai:
            def __init__(self, ptr, shape, typestr):
                self.__cuda_array_interface__ = {"shape": shape, "typestr": typestr, "data": (ptr, False),
                                                 "version": 2, "strides": None}
        dev = torch.device("cuda", torch.cuda.current_device())
        # accumulated quantities are fp64 in both precisions (rtmi_device_view); n and its gradient are of the dtype
        out = {name: torch.as_tensor(_Cai(getattr(v, name), (self.R,), "<f8" if name in _ACC else ts), device=dev)
               for name in ("x", "y", "theta", "n", "gx", "gy", "dist_sim", "dist_real", "T")}
        out["istep"] = torch.as_tensor(_Cai(v.istep, (self.R,), "<i4"), device=dev)
        if v.perm:   # sort_rays: slot k of every tensor here is the caller's ray perm[k]
            out["perm"] = torch.as_tensor(_Cai(v.perm, (self.R,), "<i4"), device=dev)
        if v.s_ray:
            out["s_ray"] = torch.as_tensor(_Cai(v.s_ray, (int(v.rec_rows), 6, self.R), ts), device=dev)
        if v.n_ray:
            out["n_ray"] = torch.as_tensor(_Cai(v.n_ray, (int(v.rec_rows), self.R), ts), device=dev)
        return out

    def close(self):
        if self._h:
            lib().rtmi_batch_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Shard:
    """One trazar() call's rays over several GPUs of this node from ONE process (rtmi_shard, include/rtmi.h): rays dealt to
    `devices` round-robin, each device builds the field and runs its rays, the read-back gathers to devices[0] device to device
    (transport "auto" | "rccl" | "copy").  Listing one device several times rehearses the split on a single GPU (copies)."""

    def __init__(self, scenario, method, step, max_size, box, gamma, thetas, x0, y0, devices, limits=None, delta=DELTA, dtype=F64,
                 record_stride=1, rec_rows=0, transport="auto", gamma_step=None, launch_mode="auto", reference_order=False,
                 keep_n_ray=False):
        sc = SCENARIOS[scenario] if isinstance(scenario, str) else scenario
        if limits is None:
            limits = constants(USER_CHOICE[scenario])[5:9]
        th = np.ascontiguousarray(thetas, dtype=np.float64)
        self.R = len(th)
        x0 = np.ascontiguousarray(np.broadcast_to(np.asarray(x0, dtype=np.float64), (self.R,)))
        y0 = np.ascontiguousarray(np.broadcast_to(np.asarray(y0, dtype=np.float64), (self.R,)))
        p = Params()
        p.method = _method_id(method); p.dtype = dtype
        p.gamma = float(gamma); p.gamma_step = float(gamma if gamma_step is None else gamma_step)
        p.step = float(step); p.max_size = int(max_size); p.record_stride = int(record_stride); p.rec_rows = int(rec_rows)
        for i in range(4):
            p.box[i] = float(box[i])
        p.launch_mode = LAUNCH_MODES[launch_mode] if isinstance(launch_mode, str) else int(launch_mode)
        p.no_n_ray = int(not keep_n_ray)
        p.reference_order = ORDERS[reference_order] if isinstance(reference_order, str) else int(reference_order)
        dev = np.ascontiguousarray(devices, dtype=np.int32)
        self.devices = [int(d) for d in dev]
        self._h = C.c_void_p()
        check(lib().rtmi_shard_create(sc.code, *[float(v) for v in limits], float(delta), C.byref(p), self.R, dptr(x0), dptr(y0), dptr(th),
                                      dev.ctypes.data_as(_lib._ip), len(dev), {"auto": 0, "rccl": 1, "copy": 2}[transport],
                                      C.byref(self._h)))

    def run(self):
        check(lib().rtmi_shard_run(self._h))

    def reset(self):
        check(lib().rtmi_shard_reset(self._h))

    def d_ray(self):
        d = np.empty((3, self.R))
        check(lib().rtmi_shard_read_d_ray(self._h, dptr(d)))
        return d

    def final(self):
        out = np.empty((9, self.R))
        check(lib().rtmi_shard_read_final(self._h, dptr(out)))
        return out

    def rows(self, row0, nrows, every=1):
        s = np.empty((nrows, 6, self.R))
        check(lib().rtmi_shard_read_rows(self._h, int(row0), int(nrows), int(every), dptr(s)))
        return s

    def info(self):
        st = _lib.ShardStats()
        check(lib().rtmi_shard_info(self._h, C.byref(st)))
        out = {k: getattr(st, k) for k, _ in _lib.ShardStats._fields_ if k != "reserved_"}
        out["transport"] = {1: "rccl", 2: "copy"}.get(out["transport"], out["transport"])
        return out

    def close(self):
        if self._h:
            lib().rtmi_shard_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def device_sincos(x):
    """The library's libm-identical fp64 sin and cos (rtmi_debug_sincos), evaluated on the device -> (sin, cos)."""
    x = np.ascontiguousarray(x, dtype=np.float64)
    s = np.empty_like(x); c = np.empty_like(x)
    check(lib().rtmi_debug_sincos(x.size, dptr(x), dptr(s), dptr(c)))
    return s, c


_ACC = ("x", "y", "theta", "dist_sim", "dist_real", "T")


def launch_is_coherent(x0, y0, theta, group=64):
    """True when consecutive rays already travel together (sorted fans do): within most groups of `group`
    consecutive rays the launch points coincide to a fraction of a cell and the angles span no more than a few
    times the batch's mean angular spacing per group.  Used by sort_rays="auto"."""
    th = np.asarray(theta, dtype=np.float64)
    R = len(th)
    if R < 2 * group:
        return True
    n = (R // group) * group
    g = th[:n].reshape(-1, group)
    with np.errstate(invalid="ignore"):
        spread = np.nanmax(g, axis=1) - np.nanmin(g, axis=1)
        total = np.nanmax(th) - np.nanmin(th)
    fair = 8.0 * total * group / R + 1e-12
    xs = np.asarray(x0, dtype=np.float64)[:n].reshape(-1, group)
    ys = np.asarray(y0, dtype=np.float64)[:n].reshape(-1, group)
    same_origin = (np.ptp(xs, axis=1) < DELTA) & (np.ptp(ys, axis=1) < DELTA)
    return bool(np.mean((spread <= fair) & same_origin) > 0.75)


def max_rows(user_choice, step, divisor):
    """max_size of trazar (:796-799)."""
    c = constants(user_choice)
    return N * divisor if c[10] else int(np.ceil(c[4] / step) + 1)


def snell_angles(s_ray, d_ray, theta_v):
    """Interface exit angles (:896-919) per ray, degrees: (angsim, angreal) -- the simulated outward angle over
    the second-to-last 5 % of the trajectory and the Snell / reflection angle it should have."""
    R = s_ray.shape[2]
    angsim, angreal = np.zeros(R), np.zeros(R)
    for k in range(R):
        i = int(d_ray[2, k])
        th = theta_v[k]
        if th < np.pi / 4:
            angreal[k] = 90 - 180 * th / np.pi
        elif th == np.pi / 4:
            angreal[k] = 0
        else:
            angreal[k] = 180 * np.arcsin(np.sqrt(2) * np.sin(np.pi / 2 - th)) / np.pi
        a, b = int(9.5 * i / 10), int(9 * i / 10)
        distx = s_ray[a, 0, k] - s_ray[b, 0, k]
        disty = s_ray[a, 1, k] - s_ray[b, 1, k]
        with np.errstate(divide="ignore", invalid="ignore"):
            angsim[k] = 180 * np.arctan(np.abs(distx / disty)) / np.pi
    return angsim, angreal


def snell_errors(s_ray, d_ray, theta_v):
    """Host restatement of the exit-angle error (:918); the library evaluates it on the device (rtmi_metric)."""
    angsim, angreal = snell_angles(s_ray, d_ray, theta_v)
    return np.abs(angsim - angreal)


def _format_num(num):
    """The reference's column formatter (:929-943)."""
    if num < 0:
        return "{: >10.8f}".format(num) if abs(num) < 10 else "{: >10.7f}".format(num)
    return "{: >10.9f}".format(num) if num < 10 else "{: >10.8f}".format(num)


def closure_error(s_ray):
    """Fisheye closure error in % of 2*pi (:956, :1393)."""
    return 100 * np.linalg.norm(np.array([1, 0]) - s_ray[-1, 0:2, 0]) / (2 * np.pi)


def moment_cv(s_ray, ray_count=None):
    """Mean coefficient of variation (%) of p_x over rays 1..R-2 (:1354-1360, :1398-1402)."""
    ray_count = s_ray.shape[2] if ray_count is None else ray_count
    cvs = np.zeros(ray_count - 2)
    for i in range(1, ray_count - 1):
        masked = np.ma.masked_equal(s_ray[:, 2, i], 0).compressed()
        cvs[i - 1] = 100 * np.std(masked) / np.mean(masked)
    return np.mean(cvs)


def trazar_plan(user_choice, step, divisor, thetas=None, starts=None, box=None, gamma=None, max_size=None, record="full"):
    """What trazar's preamble settles before any ray moves (RT_bench.py:793-804): the launch conditions, the box, gamma,
    max_size and the record stride, from the preset of `user_choice` and the keyword overrides.  Needs no device."""
    g, ray_count, theta_v, pos_x, s, limx_i, limx_s, limy_i, limy_s, op_if, op_fish, _, _ = constants(user_choice)
    if thetas is not None:
        theta_v = np.asarray(thetas, dtype=np.float64)
        ray_count = len(theta_v)
    if starts is not None:
        st = np.asarray(starts, dtype=np.float64)
        x0, y0 = (st[0], st[1]) if st.ndim == 1 else (st[:, 0], st[:, 1])
    elif op_fish:
        x0, y0 = float(pos_x[0]), float(pos_x[1])            # :810
    else:
        px = np.asarray(pos_x, dtype=np.float64)
        x0 = px[:ray_count] if len(px) >= ray_count else np.full(ray_count, px[0])
        y0 = -2.0                                            # :812
    if max_size is None:
        max_size = N * divisor if op_fish else int(np.ceil(s / step) + 1)   # :796-799
    stride = 0 if record is None else (1 if record == "full" else int(record))
    return dict(ray_count=int(ray_count), theta_v=np.asarray(theta_v, dtype=np.float64)[:ray_count], x0=x0, y0=y0,
                box=(limx_i, limx_s, limy_i, limy_s) if box is None else box, gamma=g if gamma is None else gamma,
                max_size=int(max_size), stride=stride, rec_rows=(int(max_size) + stride - 1) // stride if stride else 0,
                op_interface=bool(op_if), op_fisheye=bool(op_fish))


def trazar(selected_func, z, grd, show, step, divisor, user_choice, *, thetas=None, starts=None, box=None,
           gamma=None, max_size=None, record="full", return_batch=False, launch_mode="auto", reference_order=False,
           read_rows=True):
    """RT_bench.py:766-948 on the GPU.  Positional arguments and the returned
    (s_ray[max_size,6,R], d_ray[3,R], compute_times[R], errors[R]) are the reference's.

    Keyword extensions for synthetic batches: thetas / starts ((R,2) or (2,)) / box / gamma / max_size replace
    the preset of `user_choice`; record = "full" (reference layout), an int stride, or None (s_ray is None).
    compute_times holds the device propagation time split evenly over rays, so np.sum(compute_times) is the
    quantity the reference's benchmark reads (:1526).  launch_mode: "auto" (rtmi_params' default: the library picks the
    schedule), "plain", "refill", "sliced" or the rtmi_launch_mode integer -- same bits in all of them.  reference_order=True:
    op1/2/6/8 too step in the reference's own operation order (rtmi_params.reference_order; op7 always does): they then return
    the oracle's bits (the reference's, within 1 ulp where numpy's scalar pow(x, 2) is not x*x), at about a third of the speed.
    read_rows=False (with return_batch=True): leave the recorded rows on the device (s_ray is returned as None; take them
    from Batch.device_tensors() / Batch.rows()).
    """
    pl = trazar_plan(user_choice, step, divisor, thetas, starts, box, gamma, max_size, record)
    fld = _field_of(z, grd)
    ray_count, theta_v, stride = pl["ray_count"], pl["theta_v"], pl["stride"]
    b = Batch(fld, selected_func, step, pl["max_size"], pl["box"], pl["gamma"], theta_v, pl["x0"], pl["y0"], record_stride=stride,
              sort_rays="auto", keep_n_ray=False,          # n_ray is internal to the reference's trazar (:803), never returned
              launch_mode=launch_mode, reference_order=reference_order)
    t1 = time.perf_counter()
    b.run()
    b.sync()
    t2 = time.perf_counter()
    st_ = b.stats()
    d_ray = b.d_ray()
    s_ray = b.rows() if stride and read_rows else None
    compute_times = np.full(ray_count, (st_["kernel_ms"] * 1e-3 if st_["kernel_ms"] > 0 else t2 - t1) / ray_count)
    errors = np.zeros(ray_count)
    if pl["op_interface"] and stride == 1:
        errors = b.metric("snell")          # (:896-919) evaluated on the device
        if show and s_ray is not None:   # the reference's per-ray table (:921-945)
            print_exit_table(s_ray, d_ray, errors, theta_v)
    if return_batch:
        return s_ray, d_ray, compute_times, errors, b
    b.close()
    return s_ray, d_ray, compute_times, errors


def print_exit_table(s_ray, d_ray, errors, theta_v):
    """The reference's per-ray table of the interface scenario (RT_bench.py:921-945)."""
    angsim, angreal = snell_angles(s_ray, d_ray, theta_v)
    f = _format_num
    for k in range(s_ray.shape[2]):
        i = int(d_ray[2, k])
        print(f"Coords: [ {f(s_ray[i, 0, k])} , {f(s_ray[i, 1, k])} ] | SimAng: {f(angsim[k])} | "
              f"SnellAng: {f(angreal[k])} | Err: {f(errors[k])} | InitAng: {f(theta_v[k] * 180 / np.pi)}")


def search_delta(option, z, grd, step, divisor, user_choice):
    """RT_bench.py:950-958."""
    c = constants(user_choice)
    rays, _, _, errors = trazar(option, z, grd, False, step, divisor, user_choice)
    if c[9]:
        return np.mean(errors), np.max(errors)
    if c[10]:
        return closure_error(rays)
    return rays[:, 2, :]


# --------------------------------------------------------------------------- calibration + benchmark harness
# (the reference's other consumer of trazar, restated over the GPU path)
def calibrated_delta_s(user_choice, method_choice):
    """The hard-coded calibrated DELTA_S table (RT_bench.py:1412-1455).  method_choice is the menu entry "1".."9"
    (isotropic) or "1"/"2" (anisotropic).  Returns (DELTA_S, DELTA_S_DIVISOR_FISHEYE or None)."""
    c = constants(user_choice)
    m = str(method_choice)
    if c[9] or c[11]:
        div = {"1": 38.64, "2": 38.37, "3": 2.34, "4": 2.53, "5": 2.53, "6": 2.55, "7": 30.05, "8": 2.74, "9": 2.74}[m]
        return SIGMA / div, None
    if c[10]:
        div = {"1": 4587, "2": 4556, "3": 278, "4": 300, "5": 300, "6": 303, "7": 3567, "8": 325, "9": 325}[m]
        return 2 * np.pi / div, div
    return SIGMA / (2.53 if m == "1" else 2.74), None


def delta_s_candidates(user_choice):
    """(divisors, delta_s_options) of the DELTA_S search (RT_bench.py:1302-1312)."""
    c = constants(user_choice)
    if c[9]:
        divisors = np.arange(DELTA_S_DIVISOR_UPPER_LIMIT, DELTA_S_DIVISOR_LOWER_LIMIT - DELTA_STEP, -DELTA_STEP)
        return divisors, SIGMA / divisors
    if c[10]:
        divisors = np.arange(DELTA_S_DIVISOR_FISHEYE_UPPER_LIMIT, DELTA_S_DIVISOR_FISHEYE_LOWER_LIMIT - DELTA_STEP_FISHEYE,
                             -DELTA_STEP_FISHEYE)
        return divisors, 2 * np.pi / divisors
    divisors = np.arange(DELTA_S_DIVISOR_VERT_UPPER_LIMIT, DELTA_S_DIVISOR_VERT_LOWER_LIMIT - 2 * DELTA_STEP, -DELTA_STEP)
    return divisors, SIGMA / divisors


def search_delta_sweep(option, z, grd, delta_s_options, divisors, user_choice, batched=True):
    """What executor.map(search_delta, ...) returns (RT_bench.py:1317-1318): one entry per DELTA_S candidate --
    (mean, max) exit-angle error for interface, closure % for fisheye, mean p_x CV (%) for the vert scenarios
    (the reference returns the p_x history there and reduces it at :1354-1360; the reduction runs on the device).

    batched=True runs the whole sweep as ONE candidate x ray batch (every ray carries its candidate's DELTA_S and
    max_size, rtmi_batch_set_per_ray) followed by one device metric; batched=False runs one small batch per
    candidate.  Both give the same bits per ray."""
    c = constants(user_choice)
    g, ray_count, theta_v, pos_x, s, xi, xs, yi, ys, op_if, op_fish, _, _ = c
    fld = _field_of(z, grd)
    steps = [float(v) for v in delta_s_options]
    sizes = [int(N * d) if op_fish else int(np.ceil(s / st) + 1) for st, d in zip(steps, np.asarray(divisors) + 1)]
    th = np.asarray(theta_v, dtype=np.float64)[:ray_count]
    x0, y0 = (float(pos_x[0]), float(pos_x[1])) if op_fish else (np.asarray(pos_x, float)[:ray_count], -2.0)
    stride = 0 if op_fish else 1
    kind = "snell" if op_if else ("closure" if op_fish else "px_cv")

    def reduce_(m):
        if op_if:
            return (np.mean(m), np.max(m))
        return m[0] if op_fish else np.mean(m[1:ray_count - 1])

    if not batched:
        out = []
        for st, ms in zip(steps, sizes):
            b = Batch(fld, option, st, ms, (xi, xs, yi, ys), g, th, x0, y0, record_stride=stride)
            b.run()
            out.append(reduce_(b.metric(kind)))
            b.close()
        return out
    nc = len(steps)
    b = Batch(fld, option, steps[0], max(sizes), (xi, xs, yi, ys), g, np.tile(th, nc),
              np.tile(np.broadcast_to(x0, (ray_count,)), nc), np.tile(np.broadcast_to(y0, (ray_count,)), nc),
              record_stride=stride)
    b.set_per_ray(np.repeat(steps, ray_count), np.repeat(sizes, ray_count))
    b.run()
    m = b.metric(kind).reshape(nc, ray_count)
    b.close()
    return [reduce_(m[i]) for i in range(nc)]


def find_divisor(results, divisors, user_choice, max_deviation=None):
    """The three find_index rules of the DELTA_S search (RT_bench.py:1320-1385) -> chosen divisor or None."""
    c = constants(user_choice)
    if c[9]:
        md = MAX_DEVIATION if max_deviation is None else max_deviation
        errors = [r[0] for r in results]; max_errors = [r[1] for r in results]
        if not any(e > md for e in errors) or not any(e < md for e in errors):
            return None
        for i in reversed(range(len(errors))):
            if errors[i] < md and max_errors[i] < 0.8:
                if all(e < md for e in errors[:i]) and all(e < 0.8 for e in max_errors[:i]):
                    return round(divisors[i], 2)
        return None
    if c[10]:
        md = 5 if max_deviation is None else max_deviation
        errors = list(results)
        if not any(e > md for e in errors) or not any(e < md for e in errors):
            return None
        for i in range(len(errors)):
            if errors[i] > md:
                return round(divisors[i - 1])
        return None
    md = 0.05 if max_deviation is None else max_deviation
    errors = list(results)
    if not any(e > md for e in errors) or not any(e < md for e in errors):
        return None
    for i in range(len(errors)):
        if i > 1 and errors[i] > md and all(e < md for e in errors[:i - 1]):
            return round(divisors[i - 1], 2)
    return None


def remove_outliers_iqr(data):
    """RT_bench.py:123-138."""
    data = np.asarray(data)
    q1, q3 = np.percentile(data, 25), np.percentile(data, 75)
    iqr = q3 - q1
    return data[(data >= q1 - 1.5 * iqr) & (data <= q3 + 1.5 * iqr)]


def benchmark(option, z, grd, step, divisor, user_choice, trial=100, replicas=3, max_rounds=20, **kw):
    """The reference's benchmark statistic (RT_bench.py:1516-1541) over device propagation times: rounds of
    trial*replicas runs, IQR filter, median of the last 30 %, repeat until two successive medians differ by
    < 0.5 %; returns the mean of the last two ("Completion time per scenario", seconds)."""
    _, _, _, _, b = trazar(option, z, grd, False, step, divisor, user_choice, record=None, return_batch=True, **kw)
    benchmarks = []
    try:
        for _ in range(max_rounds):
            arr = np.zeros(trial * replicas)
            for j in range(trial * replicas):
                b.reset()
                b.run()
                arr[j] = b.stats()["kernel_ms"] * 1e-3
            cleaned = remove_outliers_iqr(arr)
            benchmarks.append(np.median(cleaned[int(-0.3 * len(cleaned)):]))
            if len(benchmarks) >= 2:
                if 100 * abs(benchmarks[-1] - benchmarks[-2]) / max(benchmarks[-1], benchmarks[-2]) < 0.5:
                    break
    finally:
        b.close()
    return float(np.mean(benchmarks[-2:]))
