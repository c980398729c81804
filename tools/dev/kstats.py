#!/usr/bin/env python3
"""Dev: per-kernel totals of a rocprofv3 --kernel-trace run (sqlite results db): kstats.py <dir> [steps]"""
import glob
import sqlite3
import sys

db = glob.glob(sys.argv[1] + "/**/*results.db", recursive=True)[0]
steps = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
c = sqlite3.connect(db)
tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if "kernel_dispatch" in t][0]
ks = [t for t in tabs if "info_kernel_symbol" in t][0]
rows = list(c.execute("select s.kernel_name, count(*), sum(d.end-d.start), avg(d.end-d.start) from %s d join %s s on d.kernel_id=s.id "
                      "group by s.kernel_name order by 3 desc" % (kd, ks)))
print("total kernel time per step: %.3f ms" % (sum(r[2] for r in rows) / steps / 1e6))
for name, n, tot, avg in rows[:int(sys.argv[3]) if len(sys.argv) > 3 else 25]:
    print("%-90s n/step %6.1f  us/step %8.1f  avg %8.1f us" % (name.replace("(anonymous namespace)::", "")[:90], n / steps, tot / steps / 1e3, avg / 1e3))
