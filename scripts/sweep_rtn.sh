#!/bin/bash
# Knob sweep of the headline RTN kernel (run on the GPU box through gpurun): bash scripts/sweep_rtn.sh > gpurun_out/sweepN.log
# OQ_RTN_* are experiment knobs of rtn.hip (Tuning::from_env); the defaults are what the sweeps of round 1 selected.
cd $GRAFT_REPO_ROOT
run() { r=$(env "$@" python bench.py --no-cpu-baseline --no-extras --steps 400 --warmup 40 $MODE 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['roofline']['launch_us'], d['verified_vs_reference_digest'])"); echo "$* mode='$MODE' -> launch_us,verified = $r"; }
MODE="--layout nbits"
run OQ_RTN_WAVEK=1
run OQ_RTN_WAVEK=0
for wpb in 2 4 8; do for gpb in 1 2; do for order in 1 2; do
run OQ_RTN_WPB=$wpb OQ_RTN_GPB=$gpb OQ_RTN_ORDER=$order
done; done; done
for gk in 2 4 8 16; do run OQ_RTN_GK=$gk; done
run OQ_RTN_NT=0
run OQ_RTN_NT=3
MODE="--layout kn"
run OQ_RTN_ORDER=1
run OQ_RTN_ORDER=2 OQ_RTN_STAGE=0
