#!/bin/bash
# kernel trace of the default GPTQ pipeline (8 layers = one wave of batched factors) -> rocpd database
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_gptq3
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace -d $OUT/trace -o t -- python3 $GRAFT_REPO_ROOT/bench_gptq.py --layers 8 --no-cpu-baseline --hessian-methods "" > $OUT/trace.log 2>&1
tail -2 $OUT/trace.log
