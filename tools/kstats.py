#!/usr/bin/env python3
"""Print a rocprofv3 *_kernel_stats.csv compactly: python tools/kstats.py <dir-or-csv> [calls_per_step]"""
import csv
import glob
import sys
path = sys.argv[1]
steps = float(sys.argv[2]) if len(sys.argv) > 2 else 0
f = path if path.endswith(".csv") else sorted(glob.glob(path + "/**/*_kernel_stats.csv", recursive=True))[0]
rows = list(csv.DictReader(open(f)))
tot = 0.0
for r in rows:
    calls, avg = int(r["Calls"]), float(r["AverageNs"]) / 1e3
    if steps and calls < steps * 0.5:
        continue
    per = calls / steps if steps else 0
    tot += calls * avg
    print(f"{r['Name'][:90]:90s} calls {calls:6d} {'(%.1f/step)' % per if steps else '':12s} avg {avg:8.2f} us  min {float(r['MinNs']) / 1e3:8.2f}")
if steps:
    print(f"sum over listed kernels: {tot / steps:.1f} us per step")
