#!/usr/bin/env python3
"""Durations (us) of the launches of kernels whose name contains argv[2], in launch order, from a rocprofv3 kernel_trace.csv (argv[1])."""
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if sys.argv[2] in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
per = int(sys.argv[3]) if len(sys.argv) > 3 else 1
d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows]
for i in range(0, len(d), per):
    print(" ".join("%.1f" % v for v in d[i:i + per]))
