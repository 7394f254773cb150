#!/bin/bash
# Round 5: every profile the docs and bench.py cite, from the FINAL code (GPU box).  Results under gpurun_out/; the summarisers
# (tools/pmc_summary.py, tools/pmc_config4_summary.py) condense them into profiles/r05_*.   bash tools/r05_profiles.sh [part ...]
R=$GRAFT_REPO_ROOT
parts=${@:-headline config4 scaling sage rehearsal}
for part in $parts; do
case $part in
headline)
  bash $R/tools/profile.sh r05a || echo "profile.sh failed with $?" ;;
config4)
  bash $R/tools/r05_config4_profile.sh final > $R/gpurun_out/r05_c4_final.stdout 2>&1 || echo "config4 profile failed" ; tail -5 $R/gpurun_out/r05_c4_final.stdout ;;
scaling)
  # one-GPU measurements of the pieces a rank of the 8-GPU forms of configs[3] and configs[4] runs
  O=$R/gpurun_out/r05_scaling.jsonl; : > $O
  cd $R
  python3 tools/big_graph_run.py rmat22 64 1 bfs > /dev/null 2>&1
  for spec in "flickr 128 100 bfs" "flickr 128 100 run" "flickr 1024 50 run" "flickr 128 20 shards" "rmat22 64 5 bfs" "rmat22 64 5 run" "rmat22 512 3 run" "rmat22 64 5 shards"; do
    python3 tools/big_graph_run.py $spec >> $O 2>> $R/gpurun_out/r05_scaling.err || echo "scaling $spec failed"
  done
  cat $O ;;
sage)
  cd /tmp && export TMPDIR=/tmp
  O=$R/gpurun_out/round_r05; mkdir -p $O
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/sage_eager -- python3 $R/tools/sage_profile.py eager 60 > $O/sage_eager.log 2>&1 || echo "sage_eager failed"
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/sage_sampled_graph -- python3 $R/tools/sage_profile.py sampler 60 graph > $O/sage_sampled_graph.log 2>&1 || echo "sage_sampled failed"
  cd $R
  python3 tools/kstats.py $O/sage_eager 60 > $O/sage_eager_per_step.txt
  python3 tools/kstats.py $O/sage_sampled_graph 60 > $O/sage_sampled_graph_per_step.txt
  tail -3 $O/sage_eager_per_step.txt $O/sage_sampled_graph_per_step.txt ;;
rehearsal)
  # bench.py --gpus N WITHOUT a launcher on the command line: the parent starts its own ranks (gloo: every rank on this one card)
  cd $R
  GRAPHPOPE_BENCH_BACKEND=gloo timeout -k 10 600 python3 bench.py --gpus 2 --config 3 --steps 5 --warmup 2 > $R/gpurun_out/r05_rehearsal_config3_world2_selflaunch.json 2> $R/gpurun_out/r05_rehearsal_config3_world2_selflaunch.err || echo "rehearsal config3 failed"
  GRAPHPOPE_BENCH_BACKEND=gloo timeout -k 10 600 python3 bench.py --gpus 4 --config 1 --steps 5 --warmup 2 > $R/gpurun_out/r05_rehearsal_config1_world4_selflaunch.json 2> $R/gpurun_out/r05_rehearsal_config1_world4_selflaunch.err || echo "rehearsal config1 failed"
  head -c 600 $R/gpurun_out/r05_rehearsal_config3_world2_selflaunch.json; echo; tail -3 $R/gpurun_out/r05_rehearsal_config3_world2_selflaunch.err ;;
esac
done
echo done
