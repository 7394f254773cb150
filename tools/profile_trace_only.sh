#!/bin/bash
# Run ON THE GPU BOX.  usage: bash tools/profile_trace_only.sh <tag>   -- only the two --kernel-trace --stats passes of tools/profile.sh
tag=$1; shift
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$tag/trace -- python3 $R/bench.py --no-cpu-baseline --no-extra "$@" > $R/gpurun_out/prof_$tag.trace.log 2>&1 || exit 1
echo "trace ok"
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$tag/trace_full -- python3 $R/bench.py --no-cpu-baseline "$@" > $R/gpurun_out/prof_$tag.trace_full.log 2>&1 || exit 1
echo "full trace ok"
