#!/bin/bash
# One GPU-box visit: GPU tests, then bench.py (output under gpurun_out/).  usage: bash scripts/gpu_check.sh TAG [pytest args]
cd "$GRAFT_REPO_ROOT"
TAG=${1:-run}; shift
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -x -q "$@" > gpurun_out/gputest_$TAG.log 2>&1
rc=$?
tail -15 gpurun_out/gputest_$TAG.log
[ $rc -ne 0 ] && exit $rc
timeout -k 10 600 python bench.py --steps 20 --warmup 5 > gpurun_out/bench_$TAG.json 2> gpurun_out/bench_$TAG.err
rc=$?
tail -c 600 gpurun_out/bench_$TAG.err
python - <<PY
import json
d=json.loads(open("gpurun_out/bench_$TAG.json").read().strip().splitlines()[-1])
print(json.dumps({k:d[k] for k in ("value","ms_per_step","verified_vs_reference_digest","other_layout","batched_launch")}))
print("roofline", json.dumps(d["roofline"]))
print("seam", json.dumps(d["seam"]))
g=d["gptq"]
print("gptq", json.dumps({k:g[k] for k in ("value","seconds","verified","verification","roofline_by_method","cpu_baseline")}) if g else None)
print("cpu", json.dumps(d["cpu_baseline"]))
PY
exit $rc
