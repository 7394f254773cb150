#!/bin/bash
# usage (on the GPU box, via gpurun): bash tools_profile.sh <tag> [bench args]
# kernel-trace + stats of bench.py; CSVs land in gpurun_out/prof_<tag>/
tag=$1; shift
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$tag -- python3 $R/bench.py --no-cpu-baseline "$@" > $R/gpurun_out/prof_$tag.log 2>&1
echo "rocprof rc=$?"
