#!/bin/bash
# Round-2 sweep of the wave kernel builds: bash scripts/sweep_rtn2.sh > gpurun_out/sweep_r02.log  (on the GPU box)
cd $GRAFT_REPO_ROOT
run() { r=$(env "$@" python bench.py --no-cpu-baseline --no-extras --steps 400 --warmup 40 $MODE 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['roofline']['launch_us'], d['verified_vs_reference_digest'])"); echo "$* mode='$MODE' -> launch_us,verified = $r"; }
MODE="--layout nbits"
for rep in 1 2; do
run OQ_RTN_WPS=0
run OQ_RTN_WPS=5
done
for wpb in 1 2 4; do run OQ_RTN_WPS=5 OQ_RTN_WPB=$wpb; done
for gk in 2 4 16; do run OQ_RTN_WPS=5 OQ_RTN_GK=$gk; done
run OQ_RTN_WPS=5 OQ_RTN_ORDER=1
run OQ_RTN_WPS=5 OQ_RTN_NT=0
run OQ_RTN_WPS=5 OQ_RTN_NT=3
run OQ_RTN_WPS=5 OQ_RTN_GPB=2
MODE="--layout nbits --symmetric"
run OQ_RTN_WPS=0
run OQ_RTN_WPS=5
