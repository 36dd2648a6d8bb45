#!/usr/bin/env python
"""List the inner loops of a kernel in a device .s file: size, LDS / VALU / scratch instruction counts.
usage: tools/asm_loops.py file.s first_line last_line"""
import re, sys
F64 = r"v_\w+_f64"
lines = open(sys.argv[1]).read().split("\n")
lo, hi = int(sys.argv[2]), int(sys.argv[3])
labels = {}
for i in range(lo, hi):
    m = re.match(r"^(\.LBB\d+_\d+):", lines[i])
    if m: labels[m.group(1)] = i
loops = []
for i in range(lo, hi):
    m = re.search(r"s_cbranch_\w+\s+(\.LBB\d+_\d+)|s_branch\s+(\.LBB\d+_\d+)", lines[i])
    if m:
        t = m.group(1) or m.group(2)
        if t in labels and labels[t] < i:
            loops.append((labels[t], i))
for a, b in sorted(loops, key=lambda x: x[1] - x[0]):
    body = [l.strip() for l in lines[a:b + 1] if l.startswith("\t") and not l.strip().startswith((";", "."))]
    n = len(body)
    cnt = lambda p: sum(1 for l in body if re.match(p, l))
    if n > 400: continue
    print(f"lines {a}-{b}: {n:4d} instr | ds {cnt(r'ds_'):3d} valu {cnt(r'v_'):3d} salu {cnt(r's_(?!waitcnt|nop|cbranch|branch)'):3d} wait {cnt(r's_waitcnt'):2d} "
          f"scratch {cnt(r'scratch_'):2d} global {cnt(r'global_'):2d} f64 {cnt(F64):3d}")
