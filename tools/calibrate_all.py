#!/usr/bin/env python3
"""Run the reference's DELTA_S search (RT_bench.py:1296-1385) for every method on the GPU and print the divisor
the find_index rules pick, next to the calibrated table the reference hard-codes (:1412-1455)."""
import sys, time
sys.path.insert(0, ".")
import numpy as np
from raytracing_amd import rt_bench as rb

TABLE = {"1": {1: 38.64, 2: 38.37, 3: 2.34, 4: 2.53, 5: 2.53, 6: 2.55, 7: 30.05, 8: 2.74, 9: 2.74},
         "2": {1: 149, 2: 169, 3: 182, 4: 179, 5: 179, 6: 182, 7: 191, 8: 179, 9: 179}}   # fisheye: the 5 % set (:1444)
for choice, scen in (("1", "interface"), ("2", "fisheye")):
    F = rb.Field.build(scen)
    div, opt = rb.delta_s_candidates(choice)
    for m in range(1, 10):
        t = time.perf_counter()
        res = rb.search_delta_sweep(rb.METHODS[m], F, None, opt, div, choice)
        pick = rb.find_divisor(res, div, choice)
        print(f"{scen:10s} op{m}: picked {pick}   reference table {TABLE[choice][m]}   ({(time.perf_counter() - t) * 1e3:.0f} ms)")
