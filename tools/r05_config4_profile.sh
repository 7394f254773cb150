#!/bin/bash
# Round 5: evidence for the large-graph path (BASELINE configs[4], R-MAT scale 22 / 512 anchors, and the per-rank shapes of the
# 8-GPU form: 64 anchors of R-MAT-22, 128 anchors of the Flickr-shaped graph).  GPU box; usage: bash tools/r05_config4_profile.sh <tag>
#   kernel trace + stats, FETCH_SIZE and WRITE_SIZE in passes of their own (MI355X_MICROARCH.md: they do not fit one pass; the
#   program itself right behind `--`), per-level HIP-event times.  Results under gpurun_out/r05_c4_<tag>/.
tag=${1:-a}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r05_c4_$tag
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
python3 $R/tools/big_graph_run.py rmat22 512 3 run > $O/run_rmat22_512.json 2> $O/run_rmat22_512.err || { echo "plain run failed"; tail -5 $O/run_rmat22_512.err; exit 1; }
cat $O/run_rmat22_512.json
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 $R/tools/big_graph_run.py rmat22 512 3 run > $O/trace.log 2>&1 || { echo "trace failed"; exit 2; }
echo "trace ok"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/fetch -- python3 $R/tools/big_graph_run.py rmat22 512 2 run > $O/fetch.log 2>&1 || { echo "fetch failed"; exit 3; }
echo "fetch ok"
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/write -- python3 $R/tools/big_graph_run.py rmat22 512 2 run > $O/write.log 2>&1 || { echo "write failed"; exit 4; }
echo "write ok"
cd $R
for spec in "rmat22 512" "rmat22 64" "rmat22 128" "flickr 128" "flickr 1024"; do
  set -- $spec
  python3 tools/big_graph_run.py $1 $2 5 levels >> $O/levels.jsonl 2>> $O/levels.err || echo "levels $spec failed"
done
for spec in "rmat22 64 5 run" "rmat22 64 5 bfs" "rmat22 128 5 run" "flickr 128 50 run" "flickr 1024 50 run"; do
  python3 tools/big_graph_run.py $spec >> $O/runs.jsonl 2>> $O/runs.err || echo "run $spec failed"
done
python3 tools/finalize_shards_ab.py > $O/finalize_shards.txt 2>&1 || echo "finalize_shards_ab failed"
python3 tools/kstats.py $O/trace > $O/kstats.txt 2>&1
cat $O/kstats.txt $O/levels.jsonl $O/runs.jsonl
tail -20 $O/finalize_shards.txt
echo done
