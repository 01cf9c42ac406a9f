#!/bin/bash
set -u
cd $GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out/r03_run11
mkdir -p $OUT
timeout -k 10 600 python -m pytest tests/test_preprocessing.py -m gpu -x -q > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -12 $OUT/pytest.log | cut -c1-400
timeout -k 10 300 python scripts/quick_awq.py 2>&1 | grep -v amdgpu.ids
