#!/bin/bash
set -u
cd $GRAFT_REPO_ROOT
for flag in "--no-model-rtn --no-awq --no-calibration --no-seam" "--no-awq --no-calibration --no-seam" "--no-model-rtn --no-calibration --no-seam" ""; do
timeout -k 10 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --gptq-extra-passes corrected $flag 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read()); g=d['gptq']; print('[$flag]', g['seconds']['wall'], g['corrected']['seconds'])"
done
