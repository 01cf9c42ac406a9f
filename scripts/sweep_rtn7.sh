#!/bin/bash
# needs the attribution build: python -m onnx_quantize_amd._build --attribution (scripts/README.md)
cd $GRAFT_REPO_ROOT
run() { r=$(env "$@" python bench.py --no-cpu-baseline --no-extras --steps 600 --warmup 60 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['roofline']['launch_us'], d['verified_vs_reference_digest'])"); echo "$* -> launch_us,verified = $r"; }
for rep in 1 2; do
run OQ_RTN_GPB=1
run OQ_RTN_GPB=2
run OQ_RTN_GPB=4
run OQ_RTN_GPB=4 OQ_RTN_GK=1
run OQ_RTN_GPB=4 OQ_RTN_GK=2
run OQ_RTN_GPB=2 OQ_RTN_GK=2
run OQ_RTN_GPB=4 OQ_RTN_NT=25
done
