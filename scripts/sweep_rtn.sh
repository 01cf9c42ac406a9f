#!/bin/bash
# Knob sweep of the headline RTN kernel (run on the GPU box through gpurun): scripts/sweep_rtn.sh > gpurun_out/sweepN.log
cd $GRAFT_REPO_ROOT
MODE="--layout nbits"
run() { r=$(env "$@" python bench.py --no-cpu-baseline --no-extras --steps 400 --warmup 40 $MODE 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['roofline']['launch_us'], d['verified_vs_reference_digest'])"); echo "$* mode='$MODE' -> launch_us,verified = $r"; }
run A=0
run OQ_RTN_MINW=5
run A=0
run OQ_RTN_MINW=5
