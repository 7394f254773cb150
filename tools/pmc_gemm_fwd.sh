#!/bin/bash
# usage (GPU box): bash tools/pmc_gemm_fwd.sh <tag>   -- kernel trace + SQ / LDS / TCC counters for the stream-K forward GEMM (tools/gemm_fwd_ab.py, config-2 shape)
tag=$1
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/pmcg_$tag/trace -- python3 $R/tools/gemm_fwd_ab.py 9988 756 256 > $R/gpurun_out/pmcg_$tag.trace.log 2>&1 || echo "trace failed"
run() { name=$1; shift; rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $R/gpurun_out/pmcg_$tag/$name -- python3 $R/tools/gemm_fwd_ab.py 9988 756 256 > $R/gpurun_out/pmcg_$tag.$name.log 2>&1 || echo "$name failed"; }
run sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS
run sq2 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU
run fetch FETCH_SIZE
run write WRITE_SIZE
run grbm GRBM_GUI_ACTIVE
echo done
