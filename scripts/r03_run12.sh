#!/bin/bash
set -u
cd $GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out/r03_run12
mkdir -p $OUT
timeout -k 10 600 python -m pytest tests/test_gptq_gpu.py -m gpu -x -q -k "pipeline or hessian" > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -5 $OUT/pytest.log | cut -c1-300
for flag in "" "--no-hessian-pipeline" "" "--no-hessian-pipeline"; do
timeout -k 10 300 python bench_gptq.py --no-cpu-baseline --hessian-methods "" --extra-passes "" $flag 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$flag', d['seconds'], d['verified'], d['config']['hessian_pipeline'])"
done
