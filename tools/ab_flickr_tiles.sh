R=$GRAFT_REPO_ROOT
for round in 1 2; do for lib in base c384 c448 c480; do for k in 512 1024; do
  echo -n "$lib K=$k: "; POPE_LIB=$R/tools/_diag/lib_$lib.so python3 $R/tools/big_graph_run.py flickr $k 200 bfs 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],4))"
done; done; done
