#!/bin/bash
# rocprofv3 kernel trace of the GPTQ component run (scripts/quick_gptq.py) on the GPU box
set -u
TAG=${1:-r01}
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_gptq_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $GRAFT_REPO_ROOT/scripts/quick_gptq.py > $OUT/trace.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE SQ_WAVES SQ_INSTS_MFMA --output-format csv -d $OUT/pmc_mfma -- python3 $GRAFT_REPO_ROOT/scripts/quick_gptq.py > $OUT/pmc_mfma.log 2>&1
find $OUT -name "*.csv" | head
