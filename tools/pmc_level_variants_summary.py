#!/usr/bin/env python3
"""Condense a tools/pmc_level_variants.sh run: per variant, counters of the level kernel averaged per launch position
(12 launches per BFS), for the dense levels 3-7: python tools/pmc_level_variants_summary.py <tag> > profiles/..."""
import collections, csv, glob, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1]
names = {0: "default", 1: "non-temporal index streams", 2: "XCD-contiguous chunk ranges", 3: "both"}
for v in range(4):
    vals = collections.defaultdict(lambda: collections.defaultdict(list))
    dur = collections.defaultdict(list)
    for grp in ("tcc", "tcp"):
        fs = glob.glob(os.path.join(ROOT, "gpurun_out", f"pmclv_{tag}", f"v{v}", grp, "*", "*_counter_collection.csv"))
        if not fs:
            continue
        per_disp = collections.defaultdict(dict)
        order = []
        for r in csv.DictReader(open(fs[0])):
            if "k_bfs_level" not in r["Kernel_Name"]:
                continue
            d = int(r["Dispatch_Id"])
            if d not in per_disp:
                order.append(d)
            per_disp[d][r["Counter_Name"]] = float(r["Counter_Value"])
        for pos, d in enumerate(order):
            for c, x in per_disp[d].items():
                vals[pos % 12][c].append(x)
    print(f"variant {v} ({names[v]}): level-kernel counters per launch, dense levels (launch positions 2..6 of 12), mean over 5 BFS runs")
    for pos in range(2, 7):
        row = {c: sum(x) / len(x) for c, x in vals[pos].items()}
        if not row:
            continue
        hit = row.get("TCC_HIT_sum", 0.0) / max(row.get("TCC_REQ_sum", 1.0), 1.0)
        print(f"   level {pos + 1}: TCC_REQ {row.get('TCC_REQ_sum', 0):10.0f}  TCC_MISS {row.get('TCC_MISS_sum', 0):9.0f}  L2 hit {hit:5.3f}  "
              f"EA read requests {row.get('TCC_EA0_RDREQ_sum', 0):9.0f}  TCP->TCC reads {row.get('TCP_TCC_READ_REQ_sum', 0):10.0f}")
