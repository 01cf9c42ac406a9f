#!/bin/bash
cd $GRAFT_REPO_ROOT
run() { r=$(env "$@" python bench.py --no-cpu-baseline --no-extras --steps 600 --warmup 60 $MODE 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['roofline']['launch_us'], d['verified_vs_reference_digest'])"); echo "$* mode='$MODE' -> launch_us,verified = $r"; }
MODE="--layout nbits"
for rep in 1 2 3; do
run OQ_RTN_WPS=0 OQ_RTN_GK=8
run OQ_RTN_WPS=0 OQ_RTN_GK=4
run OQ_RTN_WPS=5 OQ_RTN_GK=8
run OQ_RTN_WPS=5 OQ_RTN_GK=4
run OQ_RTN_WPS=5 OQ_RTN_GK=4 OQ_RTN_NT=3
run OQ_RTN_WPS=5 OQ_RTN_GK=3
run OQ_RTN_WPS=5 OQ_RTN_GK=6
done
