#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/gptq_wave; mkdir -p $O
for w in 4 16 32; do
  timeout -k 10 300 python3 bench_gptq.py --no-cpu-baseline --factor-wave $w --no-overlap > $O/full_w${w}_serial.json 2> $O/full_w${w}_serial.err || { tail -5 $O/full_w${w}_serial.err; exit 1; }
done
timeout -k 10 300 python3 bench_gptq.py --no-cpu-baseline --factor-wave 0 --no-overlap > $O/full_w0_serial.json 2> $O/full_w0_serial.err || exit 1
python3 - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/gptq_wave/full_*serial.json")):
    d=json.loads(open(f).read().strip().splitlines()[-1]); print(f.split("/")[-1], d["value"], d["seconds"], d["verified"])
PY
