#!/usr/bin/env python3
"""Count the instructions on the ordinary path of a loop in hipcc's -S output: walk from the loop header to its back edge
(`s_branch <header>`), skipping every region guarded by `s_cbranch_execz L` (the rare blocks, skipped wave-uniformly).
usage: hot_path_count.py file.s [start-line-of-kernel]"""
import re
import sys
from collections import Counter

lines = open(sys.argv[1]).read().splitlines()
args = [a for a in sys.argv[2:] if a != "-v"]
lo = int(args[0]) if args else 0
start = next(i for i in range(lo, len(lines)) if "Inner Loop Header: Depth=1" in lines[i] and i > lo + 300)
label = lines[start].split(":")[0].strip()
end = next(i for i in range(start + 1, len(lines)) if re.search(r"s_c?branch\w*\s+" + re.escape(label) + r"\s*$", lines[i]))
hot, skip_to, cold = Counter(), None, 0
for l in lines[start:end + 1]:
    t = l.strip()
    m = re.match(r"([vs]_\w+|ds_\w+|global_\w+|buffer_\w+)", t)
    if skip_to is not None:
        if t.startswith(skip_to + ":"):
            skip_to = None
        elif m:
            cold += 1
        continue
    b = re.match(r"s_cbranch_execz\s+(\.\w+)", t)
    if b:
        skip_to = b.group(1)
    if m:
        hot[m.group(1)] += 1
valu = sum(c for k, c in hot.items() if k.startswith("v_"))
salu = sum(c for k, c in hot.items() if k.startswith("s_"))
cat = lambda pat: sum(c for k, c in hot.items() if re.match(pat, k))  # noqa: E731
print(f"loop {label} lines {start + 1}-{end + 1}: hot path {valu} VALU + {salu} SALU ({hot['s_nop']} s_nop, {cat('s_cbranch')} branches); "
      f"{cold} instructions in skipped rare blocks")
TRANS, F64 = r"v_(rcp|rsq|sqrt|sin|cos|exp|log)_f32", r"v_\w*f64"
print(f"   transcendental {cat(TRANS)}, packed {cat('v_pk_')}, fp64 {cat(F64)}, "
      f"cndmask {cat('v_cndmask')}, cmp {cat('v_cmp')}, mov {cat('v_mov')}")
if "-v" in sys.argv:
    for k, c in hot.most_common(40):
        print(f"   {c:5d} {k}")
