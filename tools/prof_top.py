#!/usr/bin/env python
"""Print the top kernels of a rocprofv3 results database (dev tool): python tools/prof_top.py <results.db> [steps] [rows]."""
import sqlite3
import sys

db, steps, n = sys.argv[1], float(sys.argv[2]) if len(sys.argv) > 2 else 1.0, int(sys.argv[3]) if len(sys.argv) > 3 else 12
rows = list(sqlite3.connect(db).execute("select * from top_kernels"))
print("total kernel us per step: %.1f" % (sum(r[2] for r in rows) / steps))
for r in rows[:n]:
    print("%-100s calls %5d  us/step %9.1f  avg us %9.1f  %5.1f%%" % (r[0][:100], r[1], r[2] / steps, r[3], r[4]))
