#!/bin/bash
# Hessian call length A/B: rows per X^T X call = batch-seqs x 2048 (whole-model parity pass, same box)
mkdir -p gpurun_out/bs
for bs in 32 64 128 32; do
  python3 bench_gptq.py --no-cpu-baseline --hessian-methods '' --batch-seqs $bs > gpurun_out/bs/bs_$bs.json 2> gpurun_out/bs/bs_$bs.err || exit 1
  python3 - <<PY
import json
d=json.loads(open("gpurun_out/bs/bs_$bs.json").read().strip().splitlines()[-1])
print("batch-seqs $bs", d.get("value"), d.get("seconds"), d.get("phase_seconds_sum") or d.get("phases"), flush=True)
PY
done
