set -u
O=gpurun_out/r04_a; mkdir -p $O
timeout -k 5 150 python scripts/sanity_resident.py > $O/sanity.log 2>&1 || { echo SANITY FAILED; tail -20 $O/sanity.log; exit 1; }
tail -3 $O/sanity.log
timeout -k 10 900 python -m pytest tests/test_rtn_gpu.py tests/test_api_gpu.py -m gpu -x -q > $O/pytest.log 2>&1 || { echo PYTEST FAILED; tail -30 $O/pytest.log; exit 1; }
tail -3 $O/pytest.log
timeout -k 10 300 python scripts/quick_strategies.py --json $O/strat_new.json > $O/strat_new.log 2>&1 || { echo STRAT NEW FAILED; tail $O/strat_new.log; exit 1; }
export OQ_RTN_RESIDENT=0
timeout -k 10 300 python scripts/quick_strategies.py --json $O/strat_old.json > $O/strat_old.log 2>&1 || { echo STRAT OLD FAILED; exit 1; }
unset OQ_RTN_RESIDENT
bash scripts/profile_strategies.sh old && bash scripts/profile_strategies.sh new
echo ALL DONE
