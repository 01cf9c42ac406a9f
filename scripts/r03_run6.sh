#!/bin/bash
set -u
cd $GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out/r03_run6
mkdir -p $OUT
timeout -k 10 600 python -m pytest tests/test_rtn_gpu.py -m gpu -x -q -k "list_of_weights or batched" > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -4 $OUT/pytest.log | cut -c1-300
timeout -k 10 600 python bench.py --steps 20 --warmup 5 --no-gptq --no-seam --no-calibration --no-cpu-baseline > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"; tail -c 400 $OUT/bench.err
python - <<'PY'
import json, os
d = json.load(open(os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/r03_run6/bench.json"))
print(json.dumps(d["model_rtn"])); print(d["roofline"]["frac"], d["batched_launch"])
PY
