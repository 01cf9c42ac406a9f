#!/bin/bash
set -e
mkdir -p gpurun_out/many
python3 -m pytest tests/test_gptq_gpu.py tests/test_workspace_gpu.py tests/test_calibration_driver.py -m gpu -x -q > gpurun_out/many/tests.log 2>&1 || { tail -40 gpurun_out/many/tests.log; exit 1; }
tail -3 gpurun_out/many/tests.log
python3 scripts/quick_hessian_many.py 2>&1 | tail -2
python3 scripts/quick_driver.py > gpurun_out/many/driver.log 2>&1 || { tail -20 gpurun_out/many/driver.log; exit 1; }
tail -5 gpurun_out/many/driver.log
