#!/usr/bin/env python3
"""np.arctan2 (float64) vs its restatement in oracle/rt_oracle.c (np_arctan2) and raytracing_amd/csrc/rt_exact.h (atan2_).

The reference's op1/4/7/8 take their new angle from np.arctan2 (RT_bench.py:372, :407).  On AVX512_SKX machines numpy routes
float64 arctan2 to Intel SVML's `__svml_atan28_ha` (numpy/_core/src/umath/svml/linux/avx512/svml_z0_atan2_d_ha.s, BSD-3) --
not libm: the two differ in the last bit for 7 % of arguments.  The routine's main path is ~75 instructions; its constants
(four ratio thresholds, atan of the base points 0.5, 1, 1.5, inf as hi + lo, pi hi + lo, 12 polynomial coefficients) were read
from the `__svml_datan2_ha_data_internal` object of the numpy 2.2.6 binary of the build container.  Its first arithmetic
instruction is VRCP14PD, the hardware's 14-bit reciprocal estimate: a pure function of the operand's exponent and top 16
mantissa bits, captured as a table by tools/gen_rcp14_table.c (raytracing_amd/csrc/rt_rcp14_table.h).

  python3 tools/check_np_atan2.py       # 1.6e7 argument pairs incl. 60 decades of magnitude: expects 0 mismatches (bit compare)
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    from numpy._core._multiarray_umath import __cpu_features__ as feat
    if not feat.get("AVX512_SKX"):
        print("numpy does not dispatch float64 arctan2 to SVML on this CPU: nothing to compare")
        return
    from oracle import rt_oracle as O
    rng = np.random.default_rng(2)
    N = 4_000_000
    y = np.concatenate([rng.normal(0, 1, N), rng.uniform(-0.1, 0.1, N), rng.normal(0, 1e-3, N),
                        rng.normal(0, 1, N) * 10.0 ** rng.uniform(-30, 30, N), [0.0, -0.0, 1.0, -1.0, 0.0, 1e-320, np.inf]])
    x = np.concatenate([rng.normal(0, 1, N), rng.uniform(-0.1, 0.1, N), rng.normal(0, 1, N),
                        rng.normal(0, 1, N) * 10.0 ** rng.uniform(-30, 30, N), [1.0, -1.0, 0.0, 0.0, 0.0, 1.0, -np.inf]])
    bad = O.np_arctan2(y, x).view(np.uint64) != np.arctan2(y, x).view(np.uint64)
    print(f"{x.size} argument pairs, {int(bad.sum())} mismatches")
    sys.exit(1 if bad.any() else 0)


if __name__ == "__main__":
    main()
