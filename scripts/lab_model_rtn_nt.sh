#!/bin/bash
# model-scale RTN (224 weights, fresh outputs: the writes really reach HBM) by non-temporal setting of the headline kernel
for nt in 1 3 0 2 1; do
  OQ_RTN_NT=$nt python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-gptq --no-seam --no-awq --no-calibration 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); m = d['model_rtn']
print('OQ_RTN_NT=$nt headline', d['roofline']['frac'], 'batched', d['batched_launch']['frac'], 'model one call', m['frac'], m['device_ms'], 'loop', m['per_matrix_loop_frac'], flush=True)"
done
