#!/bin/bash
# usage (GPU box): bash tools/pmc_level_variants.sh <tag>  -- L2 / L1 counters of the level kernel for the round-3 variants
# (POPE_KNOB_LEVEL_VARIANT 0 default, 1 non-temporal index streams, 2 XCD-contiguous chunk ranges, 3 both)
tag=$1
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for v in 0 1 2 3; do
  export POPE_LEVEL_VARIANT=$v
  rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum --kernel-trace --output-format csv -d $R/gpurun_out/pmclv_$tag/v$v/tcc -- python3 $R/tools/bfs_only.py > $R/gpurun_out/pmclv_$tag.v$v.tcc.log 2>&1 || echo "v$v tcc failed"
  rocprofv3 --pmc TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum --kernel-trace --output-format csv -d $R/gpurun_out/pmclv_$tag/v$v/tcp -- python3 $R/tools/bfs_only.py > $R/gpurun_out/pmclv_$tag.v$v.tcp.log 2>&1 || echo "v$v tcp failed"
done
echo done
