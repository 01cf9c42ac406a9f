set -u
O=gpurun_out/r04_f; mkdir -p $O
timeout -k 5 150 python scripts/sanity_resident.py > $O/sanity.log 2>&1 || { echo SANITY FAILED; tail -20 $O/sanity.log; exit 1; }
for s in 4096x11008 4096x4096; do
timeout -k 10 200 python scripts/lab_tensor_stamps.py build/lab/liboq_hip_stamps.so $s > $O/stamps_$s.log 2>&1 || { echo FAILED; tail $O/stamps_$s.log; exit 1; }
grep -v amdgpu $O/stamps_$s.log
done
timeout -k 10 300 python scripts/quick_strategies.py --shapes 4096x11008,4096x4096,11008x4096,256x512 --json $O/strat.json > $O/strat.log 2>&1 || { echo STRAT FAILED; tail $O/strat.log; exit 1; }
grep -v amdgpu $O/strat.log | grep "tensor" | grep int8 | cut -c1-132
timeout -k 10 900 python -m pytest tests/test_rtn_gpu.py tests/test_api_gpu.py -m gpu -q -x > $O/pytest.log 2>&1 || { echo PYTEST FAILED; tail -30 $O/pytest.log; exit 1; }
tail -2 $O/pytest.log
