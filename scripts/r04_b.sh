set -u
O=gpurun_out/r04_b; mkdir -p $O
timeout -k 5 150 python scripts/sanity_resident.py > $O/sanity.log 2>&1 || { echo SANITY FAILED; tail -20 $O/sanity.log; exit 1; }
timeout -k 10 900 python -m pytest tests/test_rtn_gpu.py tests/test_api_gpu.py tests/test_library_abi.py tests/test_seam.py tests/test_sharding_gpu.py -m gpu -q -x > $O/pytest.log 2>&1 || { echo PYTEST FAILED; tail -30 $O/pytest.log; exit 1; }
tail -2 $O/pytest.log
