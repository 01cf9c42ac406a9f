#!/bin/bash
set -u
cd $GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out/r03_run2
mkdir -p $OUT
timeout -k 10 900 python -m pytest tests/test_gptq_gpu.py -m gpu -x -q -k "corrected or block_size or deterministic or full_size or mse" > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -5 $OUT/pytest.log | cut -c1-300
timeout -k 10 300 python scripts/quick_loop.py 2>&1 | tee $OUT/loop_new.log
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o t -- python3 $GRAFT_REPO_ROOT/scripts/quick_loop.py > $OUT/trace.log 2>&1
echo "trace rc=$?"
f=$(find $OUT/trace -name "*kernel_stats.csv" | head -1); head -6 $f | cut -c1-160
