#!/usr/bin/env python
"""Reduce a rocprofv3 PC-sampling CSV to a histogram small enough to bring home.
Groups the samples by every column that describes WHERE / WHY (code object, offset, instruction,
stall / issue reason) and drops the per-sample ones (timestamps, exec masks, ids).
usage: tools/pcs_reduce.py samples.csv > hist.txt"""
import collections
import csv
import sys

DROP = ("timestamp", "exec_mask", "dispatch", "correlation", "wave", "workgroup", "chiplet", "hw_id", "thread", "queue",
        "agent", "sample", "size", "kind")
f = open(sys.argv[1], newline="")
rd = csv.reader(f)
hdr = next(rd)
keep = [i for i, h in enumerate(hdr) if not any(d in h.lower() for d in DROP)]
print("# columns:", ",".join(hdr))
print("# kept:", ",".join(hdr[i] for i in keep))
cnt = collections.Counter()
n = 0
first = []
for row in rd:
    if n < 5:
        first.append(row)
    n += 1
    cnt[tuple(row[i] for i in keep)] += 1
print("# samples:", n)
for r in first:
    print("# raw:", ",".join(r))
for k, v in cnt.most_common():
    print(v, *k, sep="\t")
