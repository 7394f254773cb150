#!/bin/bash
# Round 4: the default bench line + the multi-GPU configs rehearsed over gloo on ONE card (no RCCL with > 1 rank exists on a one-GPU box).
set -o pipefail
mkdir -p gpurun_out
python bench.py > gpurun_out/r04_bench.json 2> gpurun_out/r04_bench.err || { tail -20 gpurun_out/r04_bench.err; exit 1; }
wc -c gpurun_out/r04_bench.json
export GRAPHPOPE_BENCH_BACKEND=gloo
for cfg in 3 4; do
  for w in 2 4; do
    timeout -k 10 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node $w --master-addr 127.0.0.1 --master-port $((29500 + cfg * 10 + w)) \
        bench.py --gpus $w --config $cfg --steps 10 --warmup 2 > gpurun_out/r04_rehearsal_config${cfg}_world${w}_gloo.json 2> gpurun_out/r04_rehearsal_config${cfg}_world${w}.err \
        || { tail -20 gpurun_out/r04_rehearsal_config${cfg}_world${w}.err; exit 1; }
    wc -c gpurun_out/r04_rehearsal_config${cfg}_world${w}_gloo.json
  done
done
