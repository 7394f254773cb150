#!/bin/bash
# usage (GPU box): bash tools/collect_round_profiles.sh r03   -- the rocprofv3 --kernel-trace --stats summaries the docs cite,
# written to gpurun_out/round_<tag>/ (copy what should be judged into profiles/).
tag=$1
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/round_$tag
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
prof() { name=$1; shift; rocprofv3 --kernel-trace --stats --output-format csv -d $O/$name -- "$@" > $O/$name.log 2>&1 || echo "$name failed"; }
prof bench_headline python3 $R/bench.py --no-sage --no-extra --no-cpu-baseline
prof bench_all python3 $R/bench.py --no-cpu-baseline
prof sage_eager python3 $R/tools/sage_profile.py eager 60
prof sage_sampled_graph python3 $R/tools/sage_profile.py sampler 60 graph
cd $R
python3 tools/kstats.py $O/sage_eager 60 > $O/sage_eager_per_step.txt
python3 tools/kstats.py $O/sage_sampled_graph 60 > $O/sage_sampled_graph_per_step.txt
python3 tools/boundary_trace.py > $O/boundary_trace.txt 2>&1
echo done
