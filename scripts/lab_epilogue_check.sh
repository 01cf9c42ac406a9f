#!/bin/bash
set -e
python3 -m pytest tests/test_gptq_gpu.py tests/test_preprocessing.py tests/test_calibration_driver.py tests/test_sharding_gpu.py -m gpu -x -q 2>&1 | tail -3
python3 scripts/lab_factor_trace.py 11008 8 2>&1 | tail -1
python3 scripts/lab_factor_trace.py 4096 24 2>&1 | tail -1
python3 scripts/quick_loop.py 2>&1 | grep "ms per loop"
python3 scripts/quick_hessian_many.py 2>&1 | tail -1
python3 scripts/quick_hessian.py f16x3 4096,11008 2>&1 | grep method
