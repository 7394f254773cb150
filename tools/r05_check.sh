#!/bin/bash
# Round 5: GPU suite (runtime messages not captured: --capture=sys) + the large-graph runs.  usage: bash tools/r05_check.sh <tag> [pytest args]
tag=${1:-a}; shift
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r05_check_$tag
mkdir -p $O
cd $R
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q --capture=sys "$@" > $O/pytest.txt 2>&1; rc=$?
tail -15 $O/pytest.txt
[ $rc -ne 0 ] && { echo "pytest rc $rc"; exit $rc; }
for spec in "rmat22 512 3 run" "rmat22 512 3 levels" "rmat22 64 5 levels" "rmat22 64 5 run" "rmat22 128 5 levels" "flickr 128 50 run" "flickr 1024 50 run" "flickr 1024 5 levels" "flickr 256 5 levels"; do
  timeout -k 10 300 python3 tools/big_graph_run.py $spec >> $O/runs.jsonl 2>> $O/runs.err || echo "run $spec failed"
done
cat $O/runs.jsonl
echo done
