#!/usr/bin/env python3
"""Instruction statistics of one kernel in the gfx950 ISA dump (make -C raytracing_amd/csrc asm)."""
import re, sys
from collections import Counter
name = sys.argv[1] if len(sys.argv) > 1 else "_Z9k_advanceIdLi6ELb1EEv8BatchDevIT_Ei"
s = open("/tmp/rtmi_gfx950.s").read()
m = re.search(r"^%s:(.*?)^\s*s_endpgm" % re.escape(name), s, re.S | re.M)
body = m.group(1).split("\n")
ins = []
labels = {}
for l in body:
    t = l.strip()
    if not t or t.startswith((";", ".")) and not t.endswith(":"):
        continue
    if t.endswith(":"):
        labels[t[:-1]] = len(ins)
        continue
    ins.append(t)
print("instructions:", len(ins))
# find backward branches -> loops
loops = []
for i, t in enumerate(ins):
    mm = re.match(r"s_cbranch_\w+\s+(\S+)", t) or re.match(r"s_branch\s+(\S+)", t)
    if mm and mm.group(1) in labels and labels[mm.group(1)] <= i:
        loops.append((labels[mm.group(1)], i))
loops.sort(key=lambda ab: ab[0] - ab[1])
for a, b in loops[:4]:
    seg = ins[a:b + 1]
    c = Counter(x.split()[0] for x in seg)
    f64 = sum(v for k, v in c.items() if "f64" in k)
    print(f"loop [{a},{b}] len {b - a + 1}: f64 ops {f64}, valu {sum(v for k, v in c.items() if k.startswith('v_'))}, "
          f"salu {sum(v for k, v in c.items() if k.startswith('s_'))}, vmem {sum(v for k, v in c.items() if k.startswith(('global_', 'buffer_', 'scratch_', 'flat_')))}, "
          f"lds {sum(v for k, v in c.items() if k.startswith('ds_'))}")
    print("   ", ", ".join(f"{k}:{v}" for k, v in c.most_common(28)))
