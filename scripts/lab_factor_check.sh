#!/bin/bash
set -e
python3 -m pytest tests/test_gptq_gpu.py -m gpu -x -q 2>&1 | tail -3
python3 scripts/lab_factor_trace.py 11008 8 2>&1 | tail -1
python3 scripts/lab_factor_trace.py 4096 24 2>&1 | tail -1
OQ_HESSIAN_METHOD=1 python3 scripts/lab_factor_trace.py 11008 8 2>&1 | tail -1
OQ_HESSIAN_METHOD=1 python3 scripts/lab_factor_trace.py 4096 24 2>&1 | tail -1
