#!/bin/bash
# GPU box: per-level times of the variant libraries built by tools/ab_variants.sh.   bash tools/ab_variants_run.sh <out> "<graph K>" lib1 lib2 ...
out=$1; spec=$2; shift 2
R=$GRAFT_REPO_ROOT
mkdir -p $(dirname $R/gpurun_out/$out)
python3 $R/tools/big_graph_run.py rmat22 64 1 bfs > /dev/null 2>&1      # caches /tmp/rmat22.npy
for round in 1 2; do
  for lib in "$@"; do
    echo "== $lib ($spec) ==" >> $R/gpurun_out/$out
    POPE_LIB=$R/tools/_diag/lib_$lib.so python3 $R/tools/big_graph_run.py $spec 4 levels >> $R/gpurun_out/$out 2>/dev/null
  done
done
python3 - <<PY
import json,re
lines=open("$R/gpurun_out/$out").read().split("\n")
cur=None; res={}
for l in lines:
    m=re.match(r"== (\S+) ",l)
    if m: cur=m.group(1); continue
    if l.startswith("{"):
        d=json.loads(l); res.setdefault(cur,[]).append(d)
for k,v in res.items():
    lv=[[x["level_us"][str(i)] for i in range(1,len(x["level_us"])+1)] for x in v]
    print(f"{k:8s} sum {[round(x['bfs_levels_sum_us']) for x in v]}  levels(last run) {lv[-1]}")
PY
