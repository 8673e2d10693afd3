#!/usr/bin/env python3
"""Instruction counts of every kernel (or one function body) in a gfx950 assembly file produced by `hipcc -S --cuda-device-only`.
usage: tools/isa_count.py file.s [name-regex]   -> per kernel: vector / scalar / LDS / memory / total instructions"""
import re
import sys

src = open(sys.argv[1]).read()
pat = re.compile(sys.argv[2]) if len(sys.argv) > 2 else None
for m in re.finditer(r"^([A-Za-z_]\w*):[^\n]*\n(.*?)s_endpgm", src, re.S | re.M):
    name = m.group(1)
    if pat and not pat.search(name):
        continue
    ins = [l.strip().split()[0] for l in m.group(2).splitlines() if l.startswith("\t") and l.strip() and not l.strip().startswith((".", ";"))]
    v = sum(i.startswith("v_") for i in ins)
    s = sum(i.startswith("s_") for i in ins)
    ds = sum(i.startswith("ds_") for i in ins)
    mem = sum(i.startswith(("global_", "buffer_", "scratch_", "flat_")) for i in ins)
    print(f"{name[:90]:90s} valu {v:5d} salu {s:5d} lds {ds:4d} mem {mem:4d} total {len(ins):5d}")
