#!/bin/bash
# usage (GPU box): bash tools/pmc_sage.sh <tag>   -- HBM bytes of the SAGE gather kernels and MFMA occupancy of the GEMMs.
# One counter group per pass, --kernel-trace only (gpurun refuses --pmc together with the other trace domains).
tag=$1
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
run() { name=$1; shift; rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $R/gpurun_out/pmcsage_$tag/$name -- python3 $R/tools/sage_profile.py eager 60 > $R/gpurun_out/pmcsage_$tag.$name.log 2>&1 || echo "$name failed"; }
run fetch FETCH_SIZE
run write WRITE_SIZE
run mfma SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_WAVE_CYCLES GRBM_GUI_ACTIVE
echo done
