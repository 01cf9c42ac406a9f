#!/bin/bash
# where does the min-max roofline of the calibration object lose its rate inside bench.py?
for flags in "--no-gptq" "--no-gptq --no-model-rtn" "--no-gptq --no-seam" "--no-gptq --no-model-rtn --no-seam"; do
  python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline $flags 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$flags', d['calibration']['roofline']['launch_us'], d['calibration']['roofline']['frac'], flush=True)"
done
