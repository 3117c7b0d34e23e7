#!/usr/bin/env python3
"""Parity sweep on the GPU box: every step method on every scenario it applies to, device (librtmi through the C ABI) against
the CPU oracle, on two batches each -- a fan of 4 096 rays from the scenario's launch point and 2 048 rays with random launch
points and directions anywhere in the box.  One line per case: rays with identical step counts, largest relative difference
of the final state and of every 64th recorded row (bench.parity_relerr: per quantity, relative to that quantity's largest
magnitude in the batch), and for the reference-order methods whether the whole result is the oracle's bits.  The oracle itself is pinned to the reference by tests/test_oracle_golden.py (its "pow" build
reproduces all 31 reference trajectory fixtures bit for bit).  This is a checker run (tests/ material), not product code.

  python3 tools/parity_sweep.py [--methods 7,9] > profiles/r04_parity_sweep.txt
"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from raytracing_amd import rt_bench as rb          # noqa: E402
from oracle import rt_oracle as O                   # noqa: E402
from bench import parity_relerr                     # noqa: E402

LIM = {"interface": (-2, 20, -2, 4), "fisheye": (-1.5, 1.5, -1.5, 1.5), "vert_heterogeneous": (-2, 5, -2.5, 1),
       "anisotropy": (-2, 5, -2.5, 1)}
EXACT = (3, 4, 5, 9, 10, 11)


def rel(a, b):
    return parity_relerr(a, b)


def main():
    only = None
    if "--methods" in sys.argv:
        only = {int(v) for v in sys.argv[sys.argv.index("--methods") + 1].split(",")}
    threads = min(O.max_threads(), os.cpu_count() or 1)
    rng = np.random.default_rng(2026)
    print(f"# device vs oracle ({threads} host threads); tolerance of the north star: 1e-9 relative, step counts exactly")
    print("# op column: R = rtmi_params.reference_order 1 (op1/2/6/8 in the reference's operation order too; op7 always is); F = 2 (op7 in")
    print("#            its fused form); H = 3 (op7's reference-order step on the fused field lookup)")
    print(f"{'scenario':19s} {'op':>4s} {'batch':7s} {'rays':>5s} {'ray-steps':>10s} {'same steps':>10s} {'final':>9s} {'rows/64':>9s} {'bits':>5s}")
    worst = worst_exact = worst_fused7 = worst_hybrid7 = 0.0
    all_bits = True
    t0 = time.time()
    for scen in ("vert_heterogeneous", "fisheye", "interface", "anisotropy"):
        key = "vert_heterogeneous" if scen == "anisotropy" else scen
        F = rb.Field.build(key, LIM[key], rb.DELTA)
        OF = O.Field(key, LIM[key], rb.DELTA)
        gam = 3 if scen == "anisotropy" else 1
        methods = (10, 11) if scen == "anisotropy" else range(1, 10)
        if scen == "fisheye":
            step, ms, x0f, y0f, thf = 2 * np.pi / 303, 3040, 1.0, 0.0, np.linspace(np.pi / 4, 3 * np.pi / 4, 4096)
        elif scen == "interface":
            step, ms, x0f, y0f, thf = rb.DELTA_S, 30228, -2.0, -2.0, np.linspace(2 * np.pi / 60, np.pi / 2, 4097)[:4096]
        else:
            step, ms, x0f, y0f, thf = rb.DELTA_S, 30228, -2.0, -2.0, np.linspace(0, np.pi / 2, 4096)
        lim = LIM[scen]
        xr = rng.uniform(lim[0] + 0.05, lim[1] - 0.05, 2048); yr = rng.uniform(lim[2] + 0.05, lim[3] - 0.05, 2048)
        thr = rng.uniform(-np.pi, np.pi, 2048)
        for m in methods:
            if only is not None and m not in only:
                continue
            for tag, x0, y0, th, msz in (("fan", x0f, y0f, thf, ms), ("random", xr, yr, thr, 3000)):
                R = len(th)
                if m in (5, 9, 10, 11) and scen == "interface" and tag == "fan":
                    th, R = th[::4], len(th[::4])          # the golden-section methods on 30 000-row rays: keep the oracle's share short
                o = O.trazar(OF, m, gam, step, msz, lim, x0, y0, th, record_stride=64, nthreads=threads)
                for ref_order in ((0, 2, 3) if m == 7 else (0, 1) if m in (1, 2, 6, 8) else (0,)):
                    b = rb.Batch(F, m, step, msz, lim, gam, th, x0, y0, record_stride=64, reference_order=ref_order)
                    b.run()
                    d, fin, rows = b.d_ray(), b.final(), b.rows()
                    b.close()
                    same = d[2] == o["d_ray"][2]
                    ef = rel(fin[:, same], o["final"][:, same])
                    er = rel(rows[:, :, same], o["s_ray"][:, :, same])
                    bits = bool(np.array_equal(fin, o["final"]) and np.array_equal(rows, o["s_ray"]) and np.array_equal(d, o["d_ray"]))
                    want_bits = m in EXACT or ref_order == 1 or (m == 7 and ref_order == 0)
                    print(f"{scen:19s} {m:3d}{' RFH'[ref_order]} {tag:7s} {R:5d} {int(d[2].sum()):10d} {int(same.sum()):10d} {ef:9.1e} {er:9.1e} "
                          f"{'yes' if bits else ('NO' if want_bits else '-'):>5s}", flush=True)
                    if want_bits:
                        worst_exact = max(worst_exact, ef, er)
                        all_bits &= bits
                    elif ref_order == 2:
                        worst_fused7 = max(worst_fused7, ef, er)
                    elif ref_order == 3:
                        worst_hybrid7 = max(worst_hybrid7, ef, er)
                    else:
                        worst = max(worst, ef, er)
        F.close()
    print(f"# reference-order rows (op3/4/5/7/9/10/11 always, op1/2/6/8 with R): the oracle's bits in every case: {all_bits} "
          f"(largest difference {worst_exact:.1e}); default of op1/2/6/8: largest difference {worst:.1e}; op7 opt-ins: fused (F) {worst_fused7:.1e}, fast field (H) {worst_hybrid7:.1e}; "
          f"{time.time() - t0:.0f} s")

if __name__ == "__main__":
    main()
