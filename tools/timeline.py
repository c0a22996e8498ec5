#!/usr/bin/env python
"""Print the kernel timeline of the last full step from a rocprofv3 --kernel-trace CSV (dev tool)."""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if 'sum_kernel' in r['Kernel_Name']]
a, b = idx[-3], idx[-2]
t0 = int(rows[a + 1]['Start_Timestamp'])
for r in rows[a + 1:b + 1]:
    print("%8.1f %8.1f  q%s %s" % ((int(r['Start_Timestamp']) - t0) / 1e3,
                                   (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3, r.get('Queue_Id'),
                                   r['Kernel_Name'][:60]))
