set -u
O=gpurun_out/r04_e; mkdir -p $O
bash scripts/lab_model_rtn_trace.sh > $O/trace.log 2>&1; grep -v amdgpu $O/trace.log | cut -c1-400 | tail -6
timeout -k 10 600 python -m pytest tests/test_rtn_gpu.py tests/test_hqq.py -m gpu -q -x > $O/pytest.log 2>&1 || { echo PYTEST FAILED; tail -30 $O/pytest.log; exit 1; }
tail -2 $O/pytest.log
timeout -k 5 300 python scripts/quick_searches.py > $O/searches.log 2>&1; grep -v amdgpu $O/searches.log | grep "g=128"
