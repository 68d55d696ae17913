#!/usr/bin/env python3
"""Summarise a rocprofv3 rocpd database (ROCm 7.2 default output): per-kernel count / total / avg / min / max
(ms) and, if present, summed PMC counters per kernel.  usage: rocprof_summary.py results.db [name-filter]"""
import sqlite3
import sys

db = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else ""
con = sqlite3.connect(db)
print("%-72s %6s %12s %10s %10s %10s" % ("kernel", "calls", "total_ms", "avg_ms", "min_ms", "max_ms"))
for name, cnt, tot, avg, mn, mx in con.execute(
        "select name, count(*), sum(end-start)/1e6, avg(end-start)/1e6, min(end-start)/1e6, max(end-start)/1e6 "
        "from kernels group by name order by 3 desc"):
    if flt and flt not in name:
        continue
    short = name.replace("(anonymous namespace)::", "").split("(")[0][-72:]
    print("%-72s %6d %12.3f %10.4f %10.4f %10.4f" % (short, cnt, tot, avg, mn, mx))
try:
    rows = con.execute("select name, counter_name, sum(counter_value), count(distinct dispatch_id) from pmc_events "
                       "group by 1, 2 order by 1, 2").fetchall()
    if rows:
        print("\n%-60s %-24s %18s %6s" % ("kernel", "counter", "sum", "disp"))
        for name, c, v, d in rows:
            if flt and flt not in name:
                continue
            print("%-60s %-24s %18.1f %6d" % (name.replace("(anonymous namespace)::", "").split("(")[0][-60:], c, v, d))
except sqlite3.Error:
    pass
