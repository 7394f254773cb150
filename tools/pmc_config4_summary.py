#!/usr/bin/env python3
"""Condense a tools/r05_config4_profile.sh run (gpurun_out/r05_c4_<tag>/) into profiles/r05_config4_{kernel_stats.csv,pmc.json,levels.jsonl}.

    python3 tools/pmc_config4_summary.py <tag>

FETCH_SIZE / WRITE_SIZE are reported in KiB.  Per MI355X_MICROARCH.md (HBM) FETCH_SIZE tallies a 128-byte request at 64 bytes on gfx950:
doubled here.  tools/micro/gather_rows.hip calibrates that for THIS kernel's pattern: a random row of 16 to 128 bytes from a table beyond L2
tallies 64 bytes raw = one 128-byte line (profiles/r05_micro_gather_rows_pmc.json), so the doubled figure is the bytes that crossed the fabric."""
import collections, csv, glob, json, os, shutil, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1]
# `config3`: the run of tools/r05_config3_profile.sh (gpurun_out/r05_c3/) into profiles/r05_config3_*
cfg = "config3" if tag == "config3" else "config4"
src = os.path.join(ROOT, "gpurun_out", "r05_c3" if cfg == "config3" else f"r05_c4_{tag}")
dst = os.path.join(ROOT, "profiles")
shutil.copy(glob.glob(os.path.join(src, "trace", "*", "*_kernel_stats.csv"))[0], os.path.join(dst, f"r05_{cfg}_kernel_stats.csv"))
for name in ("levels.jsonl", "runs.jsonl", "kstats.txt"):
    if os.path.exists(os.path.join(src, name)):
        shutil.copy(os.path.join(src, name), os.path.join(dst, f"r05_{cfg}_" + name))


def per_kernel(kind):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(glob.glob(os.path.join(src, kind, "*", "*_counter_collection.csv"))[0])):
        agg[r["Kernel_Name"].split("(")[0].replace("void ", "").replace("pope::", "").strip()].append(float(r["Counter_Value"]) * 1024)
    return agg


fetch, write = per_kernel("fetch"), per_kernel("write")
out = {"source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes, the program right behind --) on `python3 tools/big_graph_run.py "
                 + ("flickr 1024 2 run` = BASELINE configs[3] on one GPU" if cfg == "config3" else f"rmat22 512 2 run` = BASELINE configs[4] on one GPU, tag {tag}"),
       "units": "bytes per launch; hbm = FETCH_SIZE x 2 (gfx950 half-count, calibrated for this access pattern by tools/micro/gather_rows.hip) + WRITE_SIZE",
       "kernels": {}}
for name, f in fetch.items():
    w = write.get(name, [0.0] * len(f))
    if name.startswith("k_bfs_level"):
        calls = 4                                             # 2 warm-up + 2 timed calls of the step, the same launches each
        n = len(f) // calls
        per = [{"launch": i + 1, "fetch_raw": f[-n + i], "write": w[-n + i], "hbm": 2 * f[-n + i] + w[-n + i]} for i in range(n)]
        out["kernels"][name] = {"launches_per_step": n, "last_step": per}
        out["k_bfs_level_hbm_bytes_per_launch"] = sum(p["hbm"] for p in per) / n
        out["k_bfs_level_hbm_bytes_densest_launch"] = max(p["hbm"] for p in per)
    else:
        out["kernels"][name] = {"fetch_raw": f[-1], "write": w[-1], "hbm": 2 * f[-1] + w[-1], "launches": len(f)}
        if name.startswith("k_finalize"):
            out["k_finalize_hbm_bytes_per_launch"] = 2 * f[-1] + w[-1]
with open(os.path.join(dst, f"r05_{cfg}_pmc.json"), "w") as fh:
    json.dump(out, fh, indent=1)
print(json.dumps({k: v for k, v in out.items() if not isinstance(v, dict)}, indent=1))
