#!/bin/bash
# Round 4: every profile the docs and bench.py cite, from the FINAL code, in one GPU call (GPU box).  Results under gpurun_out/;
# tools/pmc_summary.py / pmc_sage_summary.py / pmc_pairwise_summary.py condense them into profiles/r04_*.
R=$GRAFT_REPO_ROOT
bash $R/tools/profile.sh r04a || echo "profile.sh failed with $?"
bash $R/tools/pmc_sage.sh r04a
bash $R/tools/pmc_pairwise.sh r04a fetch write sq2
cd /tmp && export TMPDIR=/tmp
O=$R/gpurun_out/round_r04
mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/sage_eager -- python3 $R/tools/sage_profile.py eager 60 > $O/sage_eager.log 2>&1 || echo "sage_eager failed"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/sage_sampled_graph -- python3 $R/tools/sage_profile.py sampler 60 graph > $O/sage_sampled_graph.log 2>&1 || echo "sage_sampled failed"
cd $R
python3 tools/kstats.py $O/sage_eager 60 > $O/sage_eager_per_step.txt
python3 tools/kstats.py $O/sage_sampled_graph 60 > $O/sage_sampled_graph_per_step.txt
for t in 0 4; do
  echo "== fresh process, GRAPHPOPE_PREFAULT_THREADS=$t ==" >> $O/first_call_fresh_process.txt
  GRAPHPOPE_PREFAULT_THREADS=$t python3 tools/first_call_breakdown.py whole >> $O/first_call_fresh_process.txt 2>&1
done
python3 tools/pmc_summary.py r04a r04 > $O/pmc_summary.log 2>&1
python3 tools/pmc_sage_summary.py r04a r04 > $O/pmc_sage_summary.log 2>&1
python3 tools/pmc_pairwise_summary.py r04a r04 > $O/pmc_pairwise_summary.log 2>&1
mkdir -p $R/gpurun_out/r04_profiles_out && cp $R/profiles/r04_kernel_stats*.csv $R/profiles/r04_pmc_summary.json $R/profiles/r04_sage_counters.json $R/profiles/r04_pairwise_pmc.json $R/gpurun_out/r04_profiles_out/ 2>/dev/null
echo done
