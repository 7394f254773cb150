#!/bin/bash
# usage (GPU box): bash tools/pmc_pairwise.sh <tag> [counter set ...]  -- SQ / LDS counters for the pairwise call at configs[2] (tools/pairwise_time.py)
tag=$1; shift
sets=${@:-sq1 sq2 sq3 grbm}
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
run() { name=$1; shift; rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $R/gpurun_out/pmcpw_$tag/$name -- python3 $R/tools/pairwise_time.py > $R/gpurun_out/pmcpw_$tag.$name.log 2>&1 || echo "$name failed"; }
for s in $sets; do
case $s in
sq1) run sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS ;;
sq2) run sq2 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU ;;
sq3) run sq3 SQ_LDS_UNALIGNED_STALL SQ_LDS_ADDR_CONFLICT SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL ;;
grbm) run grbm GRBM_GUI_ACTIVE ;;
fetch) run fetch FETCH_SIZE ;;
write) run write WRITE_SIZE ;;
esac
done
echo done
