#!/bin/bash
set -e
python3 -m pytest tests/test_gptq_gpu.py tests/test_calibration_driver.py tests/test_sharding_gpu.py -m gpu -x -q 2>&1 | tail -3
bash scripts/lab_syrk_splits.sh 2>&1 | grep auto
python3 bench_gptq.py --no-cpu-baseline --hessian-methods '' > gpurun_out/splits_gptq.json 2> gpurun_out/splits_gptq.err
python3 -c "
import json; d=json.loads(open('gpurun_out/splits_gptq.json').read().strip().splitlines()[-1]); print(d['seconds'], d['hessian_check'] if 'hessian_check' in d else '')"
