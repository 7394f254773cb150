#!/usr/bin/env python3
"""Timeline of the kernels around the last-but-one launch of a kernel (name substring argv[2]) in a rocprofv3 kernel_trace.csv (argv[1])."""
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1]))]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if sys.argv[2] in r["Kernel_Name"]][-2]
before, after = (int(sys.argv[3]), int(sys.argv[4])) if len(sys.argv) > 4 else (4, 8)
t0 = int(rows[idx - before]["Start_Timestamp"])
for r in rows[idx - before:idx + after]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print("%-44s %8.1f -> %8.1f  (%.1f us)" % (r["Kernel_Name"][:44], (s - t0) / 1e3, (e - t0) / 1e3, (e - s) / 1e3))
