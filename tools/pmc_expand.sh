#!/bin/bash
# usage (GPU box): bash tools/pmc_expand.sh <tag>   -- SQ / cache counters for the BFS kernels
tag=$1
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
run() { name=$1; shift; rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $R/gpurun_out/pmc_$tag/$name -- python3 $R/tools/bfs_only.py > $R/gpurun_out/pmc_$tag.$name.log 2>&1 || echo "$name failed"; }
run sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_VMEM_RD
run sq2 SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM_RD SQ_WAIT_INST_LDS
run tcc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum
run tcp TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_WRITE_REQ_sum TCP_TOTAL_ACCESSES_sum
rocprofv3 -L > $R/gpurun_out/pmc_$tag.counters.txt 2>&1
echo done
