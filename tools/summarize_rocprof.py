#!/usr/bin/env python3
"""Summarise a rocprofv3 --kernel-trace results database: per-kernel calls, total/avg/min/max us and share.
usage: summarize_rocprof.py <results.db> <steps profiled (for the per-step column)>"""
import sqlite3
import sys

db, steps = sys.argv[1], float(sys.argv[2]) if len(sys.argv) > 2 else 0.0
c = sqlite3.connect(db)
rows = c.execute("select name, count(*), sum(end-start)/1e3, avg(end-start)/1e3, min(end-start)/1e3, max(end-start)/1e3 "
                 "from kernels group by name order by 3 desc").fetchall()
tot = sum(r[2] for r in rows)
# steps profiled = launches of the once-per-step optimiser kernel (the command runs warm-up, timed, synced-median and
# roofline-pass steps: counting them by hand went wrong in round 2); the argument is only a fall-back
once = [r[1] for r in rows if "adam_clip_kernel" in r[0]]
if once:
    steps = float(once[0])
steps = steps or 1.0
print("# rocprofv3 --kernel-trace --stats summary; %d kernel names, %.3f ms of kernel time, %.3f ms per step (%g steps)"
      % (len(rows), tot / 1e3, tot / 1e3 / steps, steps))
print("%-78s %7s %11s %9s %9s %9s %6s" % ("kernel", "calls", "total_us", "avg_us", "min_us", "max_us", "pct"))
for n, k, t, a, mn, mx in rows[:40]:
    n = n.replace("(anonymous namespace)::", "").replace("void ", "")
    print("%-78s %7d %11.1f %9.2f %9.2f %9.2f %6.2f" % (n[:78], k, t, a, mn, mx, 100 * t / tot))
