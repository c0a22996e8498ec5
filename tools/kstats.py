#!/usr/bin/env python
"""Print the top rows of a rocprofv3 kernel_stats.csv (dev tool): python tools/kstats.py <dir or csv> [rows] [divide-by]."""
import csv
import glob
import os
import sys

p = sys.argv[1]
f = p if p.endswith(".csv") else sorted(glob.glob(os.path.join(p, "**", "*kernel_stats.csv"), recursive=True))[0]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 30
div = float(sys.argv[3]) if len(sys.argv) > 3 else 1.0
rows = list(csv.DictReader(open(f)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("total kernel time %.3f ms (/%g = %.3f ms)" % (tot / 1e6, div, tot / 1e6 / div))
for r in rows[:n]:
    print("%-100s calls %6s total_ms %9.3f avg_us %9.2f %5.1f%%" % (r["Name"][:100], r["Calls"], float(r["TotalDurationNs"]) / 1e6,
                                                               float(r["AverageNs"]) / 1e3, 100 * float(r["TotalDurationNs"]) / tot))
