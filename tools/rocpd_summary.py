#!/usr/bin/env python
"""Summarise a rocprofv3 rocpd SQLite database (what `rocprofv3 --kernel-trace [--pmc ...]`
writes on this image) as text: per-kernel calls / total / average duration, launch
geometry and, when present, PMC counter sums per kernel.  Usage: rocpd_summary.py <db> [...]"""
import sqlite3
import sys


def summarise(path):
    db = sqlite3.connect(path)
    cur = db.cursor()
    print(f"# {path}")
    rows = cur.execute(
        "select name, count(*), sum(duration), avg(duration), min(duration), max(duration), max(grid_x), max(workgroup_x),"
        " max(lds_size), max(scratch_size), max(vgpr_count), max(accum_vgpr_count), max(sgpr_count)"
        " from kernels group by name order by sum(duration) desc").fetchall()
    print("kernel,calls,total_ns,avg_ns,min_ns,max_ns,grid_x,wg_x,lds,scratch,vgpr,agpr,sgpr")
    for r in rows:
        print(",".join(str(x) for x in r))
    try:
        pmc = cur.execute(
            "select k.name, p.counter_name, sum(p.counter_value), count(*) from pmc_events p join kernels k"
            " on p.dispatch_id = k.dispatch_id group by k.name, p.counter_name order by k.name").fetchall()
    except sqlite3.Error:
        pmc = []
        try:
            cols = [d[1] for d in cur.execute("pragma table_info('pmc_events')")]
            print("# pmc_events columns:", cols)
        except sqlite3.Error:
            pass
    if pmc:
        print("kernel,counter,sum,dispatches")
        for r in pmc:
            print(",".join(str(x) for x in r))


if __name__ == "__main__":
    for p in sys.argv[1:]:
        summarise(p)
