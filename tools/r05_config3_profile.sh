#!/bin/bash
# Round 5: the same evidence for BASELINE configs[3] on one GPU (Flickr-shaped graph, 1 024 anchors, features resident): kernel trace +
# stats, FETCH_SIZE and WRITE_SIZE in passes of their own, the program right behind `--`.  GPU box; results under gpurun_out/r05_c3/.
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r05_c3
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 $R/tools/big_graph_run.py flickr 1024 50 run > $O/trace.log 2>&1 || { echo "trace failed"; exit 2; }
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/fetch -- python3 $R/tools/big_graph_run.py flickr 1024 2 run > $O/fetch.log 2>&1 || { echo "fetch failed"; exit 3; }
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/write -- python3 $R/tools/big_graph_run.py flickr 1024 2 run > $O/write.log 2>&1 || { echo "write failed"; exit 4; }
cd $R
python3 tools/big_graph_run.py flickr 1024 50 levels > $O/levels.jsonl 2> $O/levels.err || echo "levels failed"
python3 tools/kstats.py $O/trace > $O/kstats.txt 2>&1
cat $O/kstats.txt $O/levels.jsonl
echo done
