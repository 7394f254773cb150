#!/usr/bin/env python3
"""Condense a tools/pmc_sage.sh run into profiles/<round>_sage_counters.json (python tools/pmc_sage_summary.py r01 r01):
per SAGE kernel the average duration, the HBM-side bytes (FETCH_SIZE x 2 on gfx950, WRITE_SIZE) and, for the GEMMs, the
MFMA busy fraction and the f32 MFMA op count."""
import collections, csv, glob, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def load(tag, grp):
    fs = glob.glob(os.path.join(ROOT, "gpurun_out", f"pmcsage_{tag}", grp, "*", "*_counter_collection.csv"))
    per = collections.defaultdict(lambda: collections.defaultdict(list))
    if fs:
        for r in csv.DictReader(open(fs[0])):
            per[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    dur = collections.defaultdict(list)
    fs = glob.glob(os.path.join(ROOT, "gpurun_out", f"pmcsage_{tag}", grp, "*", "*_kernel_trace.csv"))
    if fs:
        for r in csv.DictReader(open(fs[0])):
            dur[r["Kernel_Name"]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    return per, dur


def main(tag, rnd):
    fetch, dur = load(tag, "fetch")
    write, _ = load(tag, "write")
    mfma, _ = load(tag, "mfma")
    out = {"source": f"rocprofv3 --pmc, one group per pass, on tools/sage_profile.py eager 60 (bench.py's headline SAGE step, 60 training steps), tag {tag}",
           "units": "us per launch (under counter collection); bytes per launch: FETCH_SIZE KiB x 1024 x 2 (gfx950), WRITE_SIZE KiB x 1024",
           "kernels": {}}
    for name in sorted(fetch):
        if "pope::" not in name:
            continue
        short = name.split("(")[0].replace("void ", "").replace("pope::", "")
        if not dur.get(name) or not fetch[name].get("FETCH_SIZE"):
            continue
        top = sorted(dur[name])[len(dur[name]) // 2:]                      # the larger half: layer 0 launches
        f = sorted(fetch[name]["FETCH_SIZE"])[len(fetch[name]["FETCH_SIZE"]) // 2:]
        w = sorted(write[name]["WRITE_SIZE"])[len(write[name]["WRITE_SIZE"]) // 2:] if name in write else [0.0]
        e = {"launches": len(dur[name]), "us_layer0": sum(top) / len(top),
             "hbm_read_bytes_layer0": sum(f) / len(f) * 2048, "hbm_write_bytes_layer0": sum(w) / len(w) * 1024}
        e["hbm_gbs_layer0"] = (e["hbm_read_bytes_layer0"] + e["hbm_write_bytes_layer0"]) / e["us_layer0"] / 1e3
        if name in mfma and "SQ_VALU_MFMA_BUSY_CYCLES" in mfma[name]:
            m = mfma[name]
            big = sorted(range(len(m["SQ_BUSY_CYCLES"])), key=lambda i: m["SQ_WAVE_CYCLES"][i])[len(m["SQ_BUSY_CYCLES"]) // 2:]
            busy = sum(m["SQ_VALU_MFMA_BUSY_CYCLES"][i] for i in big)
            act = sum(m["GRBM_GUI_ACTIVE"][i] for i in big) if "GRBM_GUI_ACTIVE" in m else 0
            e["mfma_f32_mops_layer0"] = sum(m["SQ_INSTS_VALU_MFMA_MOPS_F32"][i] for i in big) / len(big)
            e["mfma_busy_cycles_layer0"] = busy / len(big)
            e["gpu_active_cycles_layer0"] = act / len(big) if act else None      # summed over the 8 XCDs
            if act:                                                              # busy SIMD-cycles / (cycles per XCD x 1024 SIMDs)
                e["mfma_busy_frac_layer0"] = (busy / len(big)) / ((act / len(big)) / 8 * 1024)
        out["kernels"][short] = e
    with open(os.path.join(ROOT, "profiles", f"{rnd}_sage_counters.json"), "w") as fh:
        json.dump(out, fh, indent=1)
    for k, v in out["kernels"].items():
        print(k[:50], {a: (round(b, 1) if isinstance(b, float) else b) for a, b in v.items()})


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
