#!/usr/bin/env python3
"""Condense a tools/profile.sh run (gpurun_out/prof_<tag>/) into profiles/<round>_*.{csv,json}.

    python tools/pmc_summary.py r01a r01

* profiles/<round>_kernel_stats.csv   rocprofv3 --kernel-trace --stats summary of `python bench.py --no-extra` (the headline legs;
                                      <round>_kernel_stats_all_legs.csv: every leg of the default run)
* profiles/<round>_pmc_summary.json   HBM-side bytes per launch of the dominant kernels from the FETCH_SIZE / WRITE_SIZE
                                      passes.  rocprofv3 reports both in KiB.  Corrections per MI355X_MICROARCH.md §HBM:
                                      FETCH_SIZE reads exactly 1/2 of a wide coalesced stream on gfx950 -> doubled;
                                      WRITE_SIZE is exact for 16-byte streaming stores.  The doubling is calibrated only for
                                      streaming reads (k_finalize, k_csr_sorted check out against their known byte counts);
                                      for the 32-byte gathers of k_bfs_expand it is an upper bound.
"""
import collections
import csv
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def per_kernel(path):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        agg[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    return agg


def active_mean(vals):
    top = max(vals)
    act = [v for v in vals if v > 0.1 * top]          # skip the early-exit launches after the BFS has finished
    return sum(act) / len(act), len(act), len(vals)


def main(tag, rnd):
    src = os.path.join(ROOT, "gpurun_out", f"prof_{tag}")
    dst = os.path.join(ROOT, "profiles")
    os.makedirs(dst, exist_ok=True)
    shutil.copy(glob.glob(os.path.join(src, "trace", "*", "*_kernel_stats.csv"))[0], os.path.join(dst, f"{rnd}_kernel_stats.csv"))
    full = glob.glob(os.path.join(src, "trace_full", "*", "*_kernel_stats.csv"))
    if full:
        shutil.copy(full[0], os.path.join(dst, f"{rnd}_kernel_stats_all_legs.csv"))
    fetch = per_kernel(glob.glob(os.path.join(src, "fetch", "*", "*_counter_collection.csv"))[0])
    write = per_kernel(glob.glob(os.path.join(src, "write", "*", "*_counter_collection.csv"))[0])
    out = {"source": f"rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) on `python bench.py --steps 5 --no-sage`, tag {tag}",
           "units": "bytes per launch; FETCH_SIZE KiB x 1024 x 2 (gfx950 half-count correction), WRITE_SIZE KiB x 1024",
           "kernels": {}}
    for name in fetch:
        head = name.split("(")[0].replace("void ", "").strip()          # function name incl. template arguments
        if "pope::" not in name or head.startswith("at::") or head.startswith("__amd"):
            continue
        short = head.replace("pope::", "")
        f_mean, f_act, f_all = active_mean(fetch[name])
        w_mean, _, _ = active_mean(write.get(name, [0.0]))
        f_every = sum(fetch[name]) / len(fetch[name])
        w_every = sum(write[name]) / len(write[name]) if name in write else 0.0
        out["kernels"][short] = {"fetch_kib_raw": f_mean, "write_kib_raw": w_mean, "active_launches": f_act, "launches": f_all,
                                 "hbm_bytes_per_launch": f_mean * 1024 * 2 + w_mean * 1024,
                                 "hbm_bytes_per_launch_every_launch": f_every * 1024 * 2 + w_every * 1024}
    k = out["kernels"]
    # bench.py's roofline averages over EVERY launch of the level kernel (the early-exit ones included), as rocprofv3 --stats does
    out["k_bfs_level_hbm_bytes_per_launch"] = next(v["hbm_bytes_per_launch_every_launch"] for n, v in k.items() if n.startswith("k_bfs_level"))
    out["k_bfs_level_hbm_bytes_per_active_launch"] = next(v["hbm_bytes_per_launch"] for n, v in k.items() if n.startswith("k_bfs_level"))
    out["k_finalize_hbm_bytes_per_launch"] = next(v["hbm_bytes_per_launch"] for n, v in k.items() if n.startswith("k_finalize"))
    with open(os.path.join(dst, f"{rnd}_pmc_summary.json"), "w") as fh:
        json.dump(out, fh, indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
