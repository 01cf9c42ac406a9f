#!/bin/bash
# PMC passes for the Hessian SYRK kernels (scripts/quick_hess.py), run on the GPU box through gpurun.
#   profile_hess.sh <tag> [kernel substring = gemm_tn] [K = 11008] [T = 8192]   (method: environment OQ_HESSIAN_METHOD)
set -u
TAG=${1:-r01}
NEEDLE=${2:-gemm_tn}
K=${3:-11008}
T=${4:-8192}
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_hess_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
B="python3 $GRAFT_REPO_ROOT/scripts/quick_hess.py $K $T"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- $B > $OUT/trace.log 2>&1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $OUT/pmc_lds -- $B > $OUT/pmc_lds.log 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVES SQ_INSTS_SALU --output-format csv -d $OUT/pmc_mfma -- $B > $OUT/pmc_mfma.log 2>&1
python3 $GRAFT_REPO_ROOT/scripts/summarize_pmc.py $OUT $NEEDLE
