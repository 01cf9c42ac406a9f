#!/bin/bash
# kernel trace of bench_gptq.py (few layers) into a rocpd database + the un-profiled serial / overlapped runs beside it
set -u
TAG=${1:-r02}
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_gptq2_$TAG
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout -k 10 300 python3 bench_gptq.py --layers 8 --no-cpu-baseline --no-overlap > $OUT/serial.json 2> $OUT/serial.err || exit 1
timeout -k 10 300 python3 bench_gptq.py --layers 8 --no-cpu-baseline > $OUT/overlap.json 2> $OUT/overlap.err || exit 1
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace -d $OUT/trace -o t -- python3 $GRAFT_REPO_ROOT/bench_gptq.py --layers 4 --no-cpu-baseline > $OUT/trace.log 2>&1
ls -la $OUT/trace
