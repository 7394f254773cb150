#!/bin/bash
# GPU box: the stream-K weight-gradient kernel with the plain and the XCD-aware unit deal (POPE_KNOB_STREAMK_XCD): kernel times
# (rocprofv3 --stats), HBM read bytes and L2 hits / misses (--pmc, own passes) over 60 eager SAGE steps each.
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r05_streamk_xcd; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for x in 0 1; do
  export GRAPHPOPE_STREAMK_XCD=$x
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace$x -- python3 $R/tools/sage_profile.py eager 60 > $O/trace$x.log 2>&1 || echo "trace $x failed"
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/fetch$x -- python3 $R/tools/sage_profile.py eager 60 > $O/fetch$x.log 2>&1 || echo "fetch $x failed"
  rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d $O/tcc$x -- python3 $R/tools/sage_profile.py eager 60 > $O/tcc$x.log 2>&1 || echo "tcc $x failed"
done
cd $R
python3 - <<'PY'
import csv, glob, os
O = os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/r05_streamk_xcd"
for x in (0, 1):
    line = [f"deal {x}:"]
    for f in glob.glob(f"{O}/trace{x}/**/*kernel_stats.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "k_gemm_streamk_tn" in r["Name"] or "k_streamk_tn_fixup" in r["Name"]:
                line.append(f"{r['Name'][:40]} avg {float(r['AverageNs']) / 1e3:.2f} us min {float(r['MinNs']) / 1e3:.2f}")
    for tag in ("fetch", "tcc"):
        acc = {}
        for f in glob.glob(f"{O}/{tag}{x}/**/*counter_collection.csv", recursive=True):
            for r in csv.DictReader(open(f)):
                if "k_gemm_streamk_tn" in r["Kernel_Name"]:
                    acc.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
        for k, v in acc.items():
            m = sum(v) / len(v)
            line.append(f"{k} {m * 1024 * 2 / 1e6:.1f} MB read per launch" if k == "FETCH_SIZE" else f"{k} {m / 1e6:.3f} M per launch")
    print("  ".join(line))
PY
