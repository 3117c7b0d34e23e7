#!/usr/bin/env python3
"""Summarise gpurun_out/prof_<tag>/ (tools/profile_config.sh) into profiles/<tag>_kernel_stats.csv,
profiles/<tag>_pmc_summary.txt and an entry of profiles/traffic.json.

  python3 tools/profile_summary.py <tag> [<tag> ...]

HBM bytes per advance-kernel launch = (2*FETCH_SIZE + WRITE_SIZE) * 1024: rocprofv3 reports both in KB, and on gfx950
FETCH_SIZE counts half of the bytes of a coalesced read stream (MI355X_MICROARCH.md, HBM)."""
import collections
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def counters(d):
    agg = collections.defaultdict(list)
    meta = {}
    for fn in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(fn)):
            if "k_advance" in r["Kernel_Name"] or "k_trace_refill" in r["Kernel_Name"]:   # k_advance_sliced included
                agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
                meta = {"kernel": r["Kernel_Name"].split("(")[0], "vgpr": r["VGPR_Count"], "sgpr": r["SGPR_Count"],
                        "lds": r["LDS_Block_Size"], "scratch": r["Scratch_Size"], "grid": r["Grid_Size"], "wg": r["Workgroup_Size"]}
    return {k: sum(v) / len(v) for k, v in agg.items()}, {k: len(v) for k, v in agg.items()}, meta


def main():
    tj_path = os.path.join(ROOT, "profiles", "traffic.json")
    tj = json.load(open(tj_path)) if os.path.exists(tj_path) else {}
    for tag in sys.argv[1:]:
        d = os.path.join(ROOT, "gpurun_out", "prof_" + tag)
        bench = json.loads(open(os.path.join(d, "bench.json")).read().strip().splitlines()[-1])
        cfg, roof = bench["config"], bench["roofline"]
        steps = cfg["ray_steps_per_pass_rank0"]
        wsteps = steps / 64.0
        vals, ns, meta = {}, {}, {}
        for sub in ("pmc_fetch", "pmc_write", "pmc_sq", "pmc_mix"):
            v, n, m = counters(os.path.join(d, sub))
            vals.update(v); ns.update(n); meta = m or meta
        # kernel stats from the trace pass
        stats_rows = []
        for fn in glob.glob(os.path.join(d, "trace", "**", "*kernel_stats.csv"), recursive=True):
            stats_rows = list(csv.reader(open(fn)))
        if stats_rows:
            with open(os.path.join(ROOT, "profiles", f"{tag}_kernel_stats.csv"), "w", newline="") as f:
                csv.writer(f).writerows(stats_rows)
        adv = [r for r in stats_rows[1:] if r and ("k_advance" in r[0] or "k_trace_refill" in r[0])]
        lines = [f"# {tag}: python3 bench.py {open(os.path.join(d, 'args.txt')).read().strip()}",
                 f"# workload: {cfg['workload']}",
                 f"# bench (same box, un-profiled): value {bench['value']:.4e} ray-steps/s, {bench['ms_per_step']:.3f} ms per pass, "
                 f"advance kernel {roof['kernel_ms_per_pass']:.3f} ms per pass by HIP events ({roof['launches_per_pass']} launch(es))",
                 f"# kernel: {meta.get('kernel')}  VGPR {roof.get('vgprs')} (code object; rocprofv3 VGPR_Count column: {meta.get('vgpr')})  SGPR {meta.get('sgpr')}  LDS {meta.get('lds')} B/block  "
                 f"scratch {meta.get('scratch')} B/lane  grid {meta.get('grid')} x wg {meta.get('wg')}"]
        if adv:
            hdr = stats_rows[0]
            for r in adv:
                lines.append("# rocprofv3 --kernel-trace --stats: " + ", ".join(f"{h}={v}" for h, v in zip(hdr, r)))
        lines.append(f"# ray-steps per launch {steps} = {wsteps:.0f} wave-steps (64 rays)")
        lines.append(f"{'counter':28s} {'n':>3s} {'per launch':>16s} {'per wave-step':>14s}")
        for k in sorted(vals):
            lines.append(f"{k:28s} {ns[k]:3d} {vals[k]:16.6g} {vals[k] / wsteps:14.2f}")
        # what the counters were measured ON: bench.py uses the entry only for the same kernel at the same register count
        entry = {"source": f"profiles/{tag}_pmc_summary.txt", "kernel": roof.get("kernel"), "vgprs": roof.get("vgprs")}
        if "FETCH_SIZE" in vals and "WRITE_SIZE" in vals:
            hbm = (2 * vals["FETCH_SIZE"] + vals["WRITE_SIZE"]) * 1024
            entry["hbm_bytes"] = hbm
            lines.append(f"HBM bytes per launch = (2*FETCH_SIZE + WRITE_SIZE)*1024 = {hbm:.6g}  ({hbm / max(steps, 1):.2f} B per ray-step)")
            ms = roof["kernel_ms_per_pass"]
            lines.append(f"HBM rate at the un-profiled kernel time: {hbm / ms / 1e6:.1f} GB/s = {hbm / ms / 1e6 / 8000:.3f} of 8 TB/s")
        if "SQ_INSTS_VALU" in vals:
            f64 = sum(vals.get(k, 0) for k in ("SQ_INSTS_VALU_FMA_F64", "SQ_INSTS_VALU_MUL_F64", "SQ_INSTS_VALU_ADD_F64", "SQ_INSTS_VALU_TRANS_F64"))
            entry.update(valu_insts=vals["SQ_INSTS_VALU"], valu_f64_insts=f64, salu_insts=vals.get("SQ_INSTS_SALU"),
                         valu_active_quadcycles=vals.get("SQ_ACTIVE_INST_VALU"))
            ms = roof["kernel_ms_per_pass"]
            # SIMD issue time: a 64-lane fp64 instruction holds a SIMD for 4 cycles (16 lanes/clk), any other VALU
            # instruction for 2 (SIMD-32).  SQ_ACTIVE_INST_VALU is per-WAVE activity (about one quad-cycle per instruction
            # whatever its type) and exceeds the SIMD's time when waves overlap, so it is listed but not used as a bound.
            cyc = 4.0 * f64 + 2.0 * (vals["SQ_INSTS_VALU"] - f64)
            lines.append(f"vector ALU issue time: 4 x {f64:.4g} fp64 + 2 x {vals['SQ_INSTS_VALU'] - f64:.4g} other wave-instructions = {cyc:.4g} "
                         f"SIMD-cycles = {cyc / 1024 / 2.4e9 * 1e3:.2f} ms on 1024 SIMDs at the 2.4 GHz peak clock = "
                         f"{cyc / 1024 / 2.4e9 * 1e3 / ms:.3f} of the un-profiled kernel time")
        if "GRBM_GUI_ACTIVE" in vals and adv:
            try:
                avg_ns = float(adv[0][stats_rows[0].index("AverageNs")])
                lines.append(f"effective clock = GRBM_GUI_ACTIVE / 8 / kernel time = {vals['GRBM_GUI_ACTIVE'] / 8 / avg_ns:.2f} GHz (profiled pass)")
            except Exception:
                pass
        open(os.path.join(ROOT, "profiles", f"{tag}_pmc_summary.txt"), "w").write("\n".join(lines) + "\n")
        with open(os.path.join(ROOT, "profiles", f"{tag}_bench.json"), "w") as f:
            f.write(json.dumps(bench) + "\n")
        key = f"{cfg['workload'].split(',')[0]}:{cfg['rays_rank0']}:{cfg['record']}{'+n_ray' if cfg.get('n_ray_rows') else ''}:{bench['dtype']}:{cfg['method']}"
        mode = cfg.get("launch_mode_used", cfg.get("launch_mode", "plain"))
        if mode not in ("lane", "plain"):                     # bench.py looks plain-launch entries up without a suffix
            key += ":" + mode
        tj[key] = entry
        print("\n".join(lines))
    tj["_note"] = ("HBM bytes and instruction counts per advance-kernel launch from rocprofv3 --pmc (separate passes; FETCH_SIZE/WRITE_SIZE "
                   "in KB, bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024 as MI355X_MICROARCH.md prescribes for gfx950). One entry per "
                   "bench configuration: key scenario:rays:record:dtype:method, 'source' names the summary it came from. "
                   "Written by tools/profile_summary.py from tools/profile_config.sh runs.")
    json.dump(tj, open(tj_path, "w"), indent=1)


if __name__ == "__main__":
    main()
