"""raytracing_amd -- MI355X-native ray propagation (the trazar hot path of neyuru/RayTracing's
RT_bench.py) behind the reference's Python call surface.  Kernels: raytracing_amd/csrc (HIP, gfx950);
C ABI: include/rtmi.h; host API: raytracing_amd.rt_bench."""
from . import rt_bench  # noqa: F401
from .rt_bench import (Batch, Field, constants, genZ, interpolacion, n_gradient, trazar, search_delta,  # noqa: F401
                       interface, fisheye, vert_heterogeneous, anisotropy,
                       op1, op2, op3, op4, op5, op6, op7, op8, op9, op10, op11,
                       SIGMA, DELTA, DELTA_S, F64, F32)

__version__ = "0.1.0"
