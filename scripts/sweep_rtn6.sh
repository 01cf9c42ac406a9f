#!/bin/bash
# needs the attribution build: python -m onnx_quantize_amd._build --attribution (scripts/README.md)
# attribution: the wave kernel without its parameter stores (NT bit 3), without its blob stores (bit 4), without both
cd $GRAFT_REPO_ROOT
run() { r=$(env "$@" python bench.py --no-cpu-baseline --no-extras --steps 600 --warmup 60 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['roofline']['launch_us'], d['verified_vs_reference_digest'])"); echo "$* -> launch_us,verified = $r"; }
for rep in 1 2; do
run OQ_RTN_NT=1
run OQ_RTN_NT=9
run OQ_RTN_NT=17
run OQ_RTN_NT=25
run OQ_RTN_NT=3
done
