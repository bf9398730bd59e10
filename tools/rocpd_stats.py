#!/usr/bin/env python3
"""Per-kernel duration summary (the columns of rocprofv3's kernel_stats.csv) from a rocprofv3 rocpd database:
   python tools/rocpd_stats.py gpurun_out/prof/x_results.db profiles/x_kernel_stats.csv"""
import csv
import math
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
agg = {}
for name, start, end in db.execute("select name, start, end from kernels"):
    agg.setdefault(name, []).append(end - start)
total = sum(sum(v) for v in agg.values())
rows = []
for name, v in agg.items():
    mean = sum(v) / len(v)
    sd = math.sqrt(sum((x - mean) ** 2 for x in v) / max(len(v) - 1, 1))
    rows.append((name, len(v), sum(v), round(mean, 6), round(100.0 * sum(v) / total, 2), min(v), max(v), round(sd, 6)))
rows.sort(key=lambda r: -r[2])
with open(sys.argv[2], "w", newline="") as f:
    w = csv.writer(f, quoting=csv.QUOTE_NONNUMERIC)
    w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs", "StdDev"])
    w.writerows(rows)
for r in rows[:8]:
    print("%-72s %6d %12.1f ns %6.2f %%" % (r[0][:72], r[1], r[3], r[4]))
