#!/bin/bash
# round 3, first GPU call: the GPU suite after the hygiene changes, the new bench line, a kernel trace of corrected-mode GPTQ
set -u
cd $GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out/r03_run1
mkdir -p $OUT
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest.log
timeout -k 10 600 python bench.py --steps 20 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"; tail -c 600 $OUT/bench.err
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $OUT/trace_corr -o t -- python3 $GRAFT_REPO_ROOT/bench_gptq.py --mode corrected --layers 2 --no-cpu-baseline --hessian-methods "" --extra-passes "" > $OUT/trace_corr.log 2>&1
echo "trace rc=$?"; tail -2 $OUT/trace_corr.log | cut -c1-600
find $OUT/trace_corr -name "*kernel_stats.csv" | head -1 | xargs -r head -25
