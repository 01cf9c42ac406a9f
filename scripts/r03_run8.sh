#!/bin/bash
set -u
cd $GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out/r03_run8
mkdir -p $OUT
timeout -k 10 900 python -m pytest tests/test_gptq_gpu.py -m gpu -x -q > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -5 $OUT/pytest.log | cut -c1-300
timeout -k 10 300 python scripts/quick_loop.py 2>&1 | grep -v amdgpu.ids
timeout -k 10 300 python scripts/quick_hessian.py 2>&1 | grep -v amdgpu.ids | tail -8
