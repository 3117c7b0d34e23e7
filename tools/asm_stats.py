#!/usr/bin/env python3
"""Basic-block statistics of one kernel in the gfx950 ISA dump (make -C raytracing_amd/csrc asm -> /tmp/rtmi_gfx950.s).

  python3 tools/asm_stats.py [mangled kernel name] [--dump FIRST LAST]

One line per basic block: first line in the kernel's listing, label, instruction count, VALU / fp64 / SALU counts (SALU
includes s_waitcnt, s_nop and branches, as the SQ_INSTS_SALU counter does), memory instructions, where it branches.
This is how the waterfall loops around the row stores, the vmcnt(0) waits behind them and the copy+fmac pairs of the
rotation series were found.  --dump prints the instructions of a line range of the listing."""
import re
import sys
from collections import Counter

DEFAULT = "_Z9k_advanceIdLi6ELb1ELb1ELb0EEv8BatchDevIT_Ei"      # k_advance<double, 6, ISO, LDS tile, uniform DELTA_S>


def kernel_lines(name, path="/tmp/rtmi_gfx950.s"):
    out, on = [], False
    for l in open(path):
        if l.startswith(name + ":"):
            on = True
        if on:
            out.append(l.rstrip("\n"))
            if l.startswith(".Lfunc_end"):
                break
    if not out:
        sys.exit(f"{name} not found in {path}")
    return out


def main():
    args = [a for a in sys.argv[1:]]
    dump = None
    if "--dump" in args:
        i = args.index("--dump")
        dump = (int(args[i + 1]), int(args[i + 2]))
        del args[i:i + 3]
    L = kernel_lines(args[0] if args else DEFAULT)
    if dump:
        for i in range(dump[0] - 1, min(dump[1], len(L))):
            t = L[i].split(";")[0].rstrip() if not L[i].lstrip().startswith(";") else L[i].strip()[:14]
            if t:
                print(f"{i + 1:5d} {t}")
        return
    blocks, cur = [], None
    for i, l in enumerate(L):
        t = l.strip()
        if re.match(r"^\.LBB\d+_\d+:", l) or t.startswith("; %bb."):
            cur = {"name": t.split(":")[0], "line": i + 1, "ins": []}
            blocks.append(cur)
            continue
        if cur is None or not l.startswith("\t") or t.startswith((".", ";")):
            continue
        cur["ins"].append(t.split(";")[0].strip())
    tot = Counter()
    for b in blocks:
        c = Counter(x.split()[0] for x in b["ins"])
        valu = sum(v for k, v in c.items() if k.startswith("v_"))
        f64 = sum(v for k, v in c.items() if "f64" in k)
        salu = sum(v for k, v in c.items() if k.startswith("s_"))
        mem = ",".join(f"{k}:{v}" for k, v in c.items() if k.startswith(("ds_", "global_", "buffer_", "scratch_", "s_load")))
        br = ";".join(x for x in b["ins"] if x.startswith(("s_cbranch", "s_branch", "s_swappc", "s_setpc")))
        tot.update(valu=valu, f64=f64, salu=salu, n=len(b["ins"]))
        print(f"{b['line']:5d} {b['name']:12s} n={len(b['ins']):4d} valu={valu:4d} f64={f64:4d} salu={salu:3d} {mem} -> {br}")
    print(f"total: {tot['n']} instructions, {tot['valu']} VALU ({tot['f64']} fp64), {tot['salu']} SALU (static counts)")


if __name__ == "__main__":
    main()
