#!/bin/bash
cd $GRAFT_REPO_ROOT
run() { r=$(env "$@" python bench.py --no-cpu-baseline --steps 400 --warmup 40 $MODE 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['roofline']['launch_us'], d['verified_vs_reference_digest'])"); echo "$* mode='$MODE' -> launch_us,verified = $r"; }
for MODE in "--layout nbits" ""; do
run OQ_RTN_WAVES=16 OQ_RTN_ORDER=1 OQ_RTN_STAGE=1
run OQ_RTN_WAVES=16 OQ_RTN_ORDER=1 OQ_RTN_STAGE=0
for gk in 2 4 8; do run OQ_RTN_WAVES=16 OQ_RTN_ORDER=2 OQ_RTN_GK=$gk OQ_RTN_STAGE=0; done
run OQ_RTN_WAVES=8 OQ_RTN_ORDER=2 OQ_RTN_GK=8 OQ_RTN_STAGE=0
done
