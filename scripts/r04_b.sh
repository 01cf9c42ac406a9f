set -u
O=gpurun_out/r04_b; mkdir -p $O
timeout -k 5 150 python scripts/sanity_resident.py > $O/sanity.log 2>&1 || { echo SANITY FAILED; tail -20 $O/sanity.log; exit 1; }
timeout -k 10 300 python scripts/quick_strategies.py --shapes 4096x11008,4096x4096,11008x4096,256x512 --json $O/strat.json > $O/strat.log 2>&1 || { echo STRAT FAILED; tail $O/strat.log; exit 1; }
grep -v amdgpu $O/strat.log | grep "tensor\|channel" | grep int8 | cut -c1-132
timeout -k 10 900 python -m pytest tests/test_rtn_gpu.py tests/test_api_gpu.py -m gpu -q -x > $O/pytest.log 2>&1 || { echo PYTEST FAILED; tail -30 $O/pytest.log; exit 1; }
tail -2 $O/pytest.log
