#!/bin/bash
# the factor-phase stall showed in about one whole-model pass in three: six processes, parity + corrected pass each
mkdir -p gpurun_out/stall
for i in 1 2 3 4 5 6; do
  python3 bench_gptq.py --no-cpu-baseline --hessian-methods "" --extra-passes corrected > gpurun_out/stall/run$i.json 2> gpurun_out/stall/run$i.err || { tail -20 gpurun_out/stall/run$i.err; exit 1; }
  python3 -c "
import json; d=json.loads(open('gpurun_out/stall/run$i.json').read().strip().splitlines()[-1]); print($i, d['seconds'], d['corrected']['seconds'], d['verify']['verified'] if 'verify' in d else d.get('verified'), flush=True)"
done
