#!/bin/bash
# needs the attribution build: python -m onnx_quantize_amd._build --attribution (scripts/README.md)
cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_rtn_gpu.py tests/test_seam.py -m gpu -x -q > gpurun_out/rtn16.log 2>&1 || { tail -20 gpurun_out/rtn16.log; exit 1; }
tail -2 gpurun_out/rtn16.log
run() { r=$(env "$@" python bench.py --no-cpu-baseline --no-extras --steps 600 --warmup 60 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['roofline']['launch_us'], d['verified_vs_reference_digest'])"); echo "$* -> launch_us,verified = $r"; }
for rep in 1 2 3; do
run OQ_RTN_NT=1
run OQ_RTN_NT=65
run OQ_RTN_NT=3
run OQ_RTN_NT=67
done
