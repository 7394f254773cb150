#!/bin/bash
# Run ON THE GPU BOX (through gpurun).  usage: bash tools/profile.sh <tag> [bench args]
#   kernel trace + stats, then FETCH_SIZE and WRITE_SIZE in their own passes (MI355X_MICROARCH.md: they do not fit one pass;
#   gpurun refuses --pmc combined with the trace domains other than --kernel-trace).  CSVs land in gpurun_out/prof_<tag>/.
tag=$1; shift
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
# headline legs only (configs[1] step + SAGE): the per-kernel averages of THIS summary are the ones bench.py's roofline quotes
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$tag/trace -- python3 $R/bench.py --no-cpu-baseline --no-extra "$@" > $R/gpurun_out/prof_$tag.trace.log 2>&1 || exit 1
echo "trace ok"
# every leg (boundary, PageRank, node2vec, K-means, configs[3] / [4]): the same kernels at other shapes are averaged in
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$tag/trace_full -- python3 $R/bench.py --no-cpu-baseline "$@" > $R/gpurun_out/prof_$tag.trace_full.log 2>&1 || exit 1
echo "full trace ok"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/prof_$tag/fetch -- python3 $R/bench.py --no-cpu-baseline --no-sage --no-extra --steps 5 --warmup 2 > $R/gpurun_out/prof_$tag.fetch.log 2>&1 || exit 2
echo "fetch ok"
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/prof_$tag/write -- python3 $R/bench.py --no-cpu-baseline --no-sage --no-extra --steps 5 --warmup 2 > $R/gpurun_out/prof_$tag.write.log 2>&1 || exit 3
echo "write ok"
