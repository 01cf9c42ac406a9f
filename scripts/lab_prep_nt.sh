#!/bin/bash
# A/B on one box: preparation kernels (absmax, split) with plain against non-temporal loads of X, whole-model parity pass
set -e
mkdir -p gpurun_out/nt
for v in plain nt plain nt; do
  if [ $v = nt ]; then python3 -m onnx_quantize_amd._build --define OQ_PREP_NT > /dev/null; else python3 -m onnx_quantize_amd._build > /dev/null; fi
  python3 bench_gptq.py --no-cpu-baseline --hessian-methods '' > gpurun_out/nt/$v.json 2> gpurun_out/nt/$v.err
  python3 -c "
import json; d=json.loads(open('gpurun_out/nt/$v.json').read().strip().splitlines()[-1]); print('$v', d['seconds'], flush=True)"
done
python3 -m onnx_quantize_amd._build > /dev/null
