#!/usr/bin/env python
"""Spill traffic of a kernel by loop depth, from an llvm-objdump -d listing of ONE kernel (tools/kernel_regs.sh leaves the
batch kernel in /tmp/mr_last.s): counts of scratch loads / stores and v_readlane / v_writelane (SGPR spills) per loop-nesting
depth, where a loop is a backward branch.  usage: tools/isa_depth.py file.s"""
import re, sys
ins = []  # (addr, text, target)
for l in open(sys.argv[1]):
    m = re.match(r"^\t(\S.*?)\s+// ([0-9A-F]+):", l)
    if not m: continue
    text, addr = m.group(1), int(m.group(2), 16)
    t = None
    mb = re.match(r"s_(?:cbranch_\w+|branch)\s+(\d+)", text)
    if mb:
        off = int(mb.group(1))
        if off >= 32768: off -= 65536
        t = addr + 4 + 4 * off
    ins.append((addr, text, t))
loops = [(t, a) for a, _, t in ins if t is not None and t <= a]
def depth(a): return sum(1 for lo, hi in loops if lo <= a <= hi)
pats = {"scratch_load": r"scratch_load", "scratch_store": r"scratch_store", "v_readlane": r"v_readlane", "v_writelane": r"v_writelane", "all": r"."}
tab = {}
for a, text, _ in ins:
    d = min(depth(a), 4)
    for k, p in pats.items():
        if re.match(p, text): tab[(k, d)] = tab.get((k, d), 0) + 1
print("loops (backward branches):", len(loops), " instructions:", len(ins))
print("%-14s" % "depth" + "".join("%8s" % (str(d) if d < 4 else "4+") for d in range(5)))
for k in pats:
    print("%-14s" % k + "".join("%8d" % tab.get((k, d), 0) for d in range(5)))
