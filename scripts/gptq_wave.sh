#!/bin/bash
# batched-factor pipeline against the per-input one (8 layers each), then the factor tests
cd $GRAFT_REPO_ROOT
O=gpurun_out/gptq_wave; mkdir -p $O
timeout -k 10 300 python -X faulthandler -m pytest tests/test_gptq_gpu.py tests/test_calibration_driver.py -m gpu -x -q -k "factor or streamed" > $O/tests.log 2>&1 || { tail -20 $O/tests.log; exit 1; }
tail -2 $O/tests.log
for w in 0 2 4 8; do
  timeout -k 10 300 python3 bench_gptq.py --layers 8 --no-cpu-baseline --factor-wave $w > $O/w$w.json 2> $O/w$w.err || { tail -5 $O/w$w.err; exit 1; }
done
timeout -k 10 300 python3 bench_gptq.py --layers 8 --no-cpu-baseline --factor-wave 8 --no-overlap > $O/w8_serial.json 2> $O/w8_serial.err || exit 1
timeout -k 10 300 python3 bench_gptq.py --layers 8 --no-cpu-baseline --factor-wave 4 --factor-streams 2 > $O/w4_s2.json 2> $O/w4_s2.err || exit 1
python3 - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/gptq_wave/*.json")):
    d=json.loads(open(f).read().strip().splitlines()[-1]); print(f.split("/")[-1], d["value"], d["seconds"], d["verified"])
PY
