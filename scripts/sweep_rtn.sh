#!/bin/bash
cd $GRAFT_REPO_ROOT
run() { r=$(env "$@" python bench.py --no-cpu-baseline --steps 300 --warmup 30 $MODE 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['roofline']['launch_us'], d['verified_vs_reference_digest'])"); echo "$* mode='$MODE' -> launch_us,verified = $r"; }
for MODE in "--layout nbits" ""; do
  run A=default
  run OQ_RTN_STAGE_Q=0
done
