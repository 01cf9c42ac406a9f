#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/gptq_wave; mkdir -p $O
for cfg in "4 4" "8 4" "8 2" "16 2"; do
  set -- $cfg
  timeout -k 10 300 python3 bench_gptq.py --no-cpu-baseline --factor-wave $1 --factor-streams $2 > $O/full_w$1_s$2.json 2> $O/full_w$1_s$2.err || { tail -5 $O/full_w$1_s$2.err; exit 1; }
done
timeout -k 10 300 python3 bench_gptq.py --no-cpu-baseline --factor-wave 8 --no-overlap > $O/full_w8_serial.json 2> $O/full_w8_serial.err || exit 1
python3 - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/gptq_wave/full_*.json")):
    d=json.loads(open(f).read().strip().splitlines()[-1]); print(f.split("/")[-1], d["value"], d["seconds"], d["verified"])
PY
