#!/bin/bash
set -u
cd $GRAFT_REPO_ROOT
for flag in "--extra-passes corrected" "--extra-passes corrected --no-hessian-pipeline" "--extra-passes corrected --hessian-methods auto" "--extra-passes corrected --hessian-methods f32"; do
timeout -k 10 300 python bench_gptq.py --no-cpu-baseline $flag 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('$flag', d['seconds']['wall'], (d.get('corrected') or {}).get('seconds'))"
done
