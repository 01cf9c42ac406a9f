set -u
O=gpurun_out/r04_c; mkdir -p $O
timeout -k 10 1500 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1 || { echo PYTEST FAILED; tail -40 $O/pytest.log; exit 1; }
tail -3 $O/pytest.log
timeout -k 10 900 python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err || { echo BENCH FAILED; tail -30 $O/bench.err; exit 1; }
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r04_c/bench.json').read().strip().splitlines()[-1])
r=d['roofline']; print('headline', d['value'], r['frac'], r['launch_us'], 'streamed', r.get('frac_outputs_streamed'), r.get('launch_us_outputs_streamed'))
print('strategies', json.dumps(d.get('strategies'))[-700:])
print('searches', json.dumps(d.get('searches'))[:400])
print('model_rtn', d['model_rtn']['frac'], d['model_rtn']['device_ms'], d['model_rtn']['equals_single_matrix_outputs'])
print('other_layout', d['other_layout'])
g=d['gptq']; print('gptq', g['value'], g['seconds'] if 'seconds' in g else '', g.get('verification', {}).get('verified'))
print('corrected', json.dumps(g.get('corrected'))[:1500])
PY
