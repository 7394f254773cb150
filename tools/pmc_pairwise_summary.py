#!/usr/bin/env python3
"""Condense a tools/pmc_pairwise.sh run (fetch / write / sq2 / grbm passes) into profiles/<round>_pairwise_pmc.json:
    python tools/pmc_pairwise_summary.py r04a r04
per kernel of the configs[2] call the average duration under counter collection, the HBM-side bytes per launch (FETCH_SIZE KiB x
1024 x 2 on gfx950, WRITE_SIZE KiB x 1024; MI355X_MICROARCH.md HBM section) and, for the tile kernel, the MFMA / LDS counters."""
import collections, csv, glob, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
N, D, K, F = 89250, 128, 256, 500
ALGO = {"k_pairwise_persistent": (4 * (N * D + K * D), 4 * N * K), "k_copy_features<2>": (4 * N * F, 4 * N * F),
        "k_minmax_apply": (4 * N * K, 4 * N * K)}


def load(tag, grp):
    per = collections.defaultdict(lambda: collections.defaultdict(list))
    dur = collections.defaultdict(list)
    for f in glob.glob(os.path.join(ROOT, "gpurun_out", f"pmcpw_{tag}", grp, "*", "*_counter_collection.csv")):
        for r in csv.DictReader(open(f)):
            per[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for f in glob.glob(os.path.join(ROOT, "gpurun_out", f"pmcpw_{tag}", grp, "*", "*_kernel_trace.csv")):
        for r in csv.DictReader(open(f)):
            dur[r["Kernel_Name"]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    return per, dur


def short(name):
    return name.split("(")[0].replace("void ", "").replace("pope::", "")


def main(tag, rnd):
    fetch, dur = load(tag, "fetch")
    write, _ = load(tag, "write")
    sq2, _ = load(tag, "sq2")
    out = {"source": f"rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE / SQ counters (separate passes, kernels serialised by the counter collection) on "
                     f"tools/pairwise_time.py, configs[2]; tools/pmc_pairwise.sh {tag} fetch write sq2",
           "units": "bytes per launch: FETCH_SIZE KiB x 1024 x 2 (gfx950 half-count correction), WRITE_SIZE KiB x 1024", "kernels": {}}
    for name in sorted(fetch):
        s = short(name)
        if not any(s.startswith(k) for k in ("k_pairwise_persistent", "k_copy_features", "k_minmax_apply", "k_minmax_finish")):
            continue
        mean = lambda v: sum(v) / len(v) if v else 0.0
        e = {"launches": len(dur.get(name, [])), "us_under_counters": mean(dur.get(name, [])),
             "hbm_read_bytes": mean(fetch[name].get("FETCH_SIZE", [])) * 2048, "hbm_write_bytes": mean(write.get(name, {}).get("WRITE_SIZE", [])) * 1024}
        if s in ALGO:
            e["algorithmic_read_bytes"], e["algorithmic_write_bytes"] = ALGO[s]
        for c in ("SQ_VALU_MFMA_BUSY_CYCLES", "SQ_INSTS_VALU_MFMA_MOPS_F32", "SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE"):
            if name in sq2 and c in sq2[name] and s.startswith("k_pairwise_persistent"):
                e[c] = mean(sq2[name][c])
        out["kernels"][s] = e
    with open(os.path.join(ROOT, "profiles", f"{rnd}_pairwise_pmc.json"), "w") as fh:
        json.dump(out, fh, indent=1)
    for k, v in out["kernels"].items():
        print(k, {a: (round(b, 1) if isinstance(b, float) else b) for a, b in v.items()})


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
