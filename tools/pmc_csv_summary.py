#!/usr/bin/env python3
"""Summarise rocprofv3 CSVs under a directory: per kernel, average duration (kernel trace) and average counter values.
usage: python tools/pmc_csv_summary.py gpurun_out/pmcg_<tag> [kernel-name substring ...]"""
import csv, glob, os, sys, collections
root = sys.argv[1]
filt = sys.argv[2:]
def short(n): return n.split("(")[0][:60]
for sub in sorted(os.listdir(root)):
    d = os.path.join(root, sub)
    dur = collections.defaultdict(list)
    for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            dur[short(r["Kernel_Name"])].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    cnt = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            cnt[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
    print(f"== {sub}")
    for k in sorted(dur, key=lambda k: -sum(dur[k])):
        if filt and not any(s in k for s in filt): continue
        v = dur[k]
        line = f"  {k:58s} n={len(v):4d} avg={sum(v)/len(v):9.2f} us"
        for c, vals in sorted(cnt.get(k, {}).items()):
            line += f"  {c}={sum(vals)/len(vals):.4g}"
        print(line)
