#!/bin/bash
cd $GRAFT_REPO_ROOT
run() { r=$(env "$@" python bench.py --no-cpu-baseline --steps 300 --warmup 30 $MODE 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['roofline']['launch_us'], d['verified_vs_reference_digest'])"); echo "$* mode='$MODE' -> launch_us,verified = $r"; }
for MODE in "" "--layout nbits" "--symmetric"; do
  run A=default
  run OQ_RTN_ORDER=0
  run OQ_RTN_STREAM=1
  run OQ_RTN_STAGE=0
done
