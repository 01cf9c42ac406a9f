#!/bin/bash
set -u
cd $GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out/r03_final
mkdir -p $OUT
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest.log | cut -c1-300
timeout -k 10 100 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke rc=$?"; tail -2 $OUT/smoke.log
S0=$(date +%s); timeout -k 10 900 python bench.py --steps 20 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$? in $(( $(date +%s) - S0 )) s"
python - <<'PY'
import json, os
d = json.load(open(os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/r03_final/bench.json"))
g = d["gptq"]
print("headline", d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["launch_us_p10_p50_p90"], "other", d["other_layout"], "batched", d["batched_launch"]["frac"])
m = d["model_rtn"]; print("model_rtn", m["frac"], m["device_ms"], m["per_matrix_loop_frac"], m["small_matrices"])
c = d["calibration"]; print("calib", c["value"], c["roofline"]["frac"], c["verified"], c["cpu_baseline"]["value"])
print("awq", d["awq"])
print("gptq", g["value"], g["seconds"], g["verified"], g["config"]["hessian_pipeline"])
print("corrected", g["corrected"]["value"], g["corrected"]["seconds"], g["corrected"]["verified"], [s["ratio"] for s in g["corrected"]["verification"]])
print("by method", g["wall_by_hessian_method"]["f32"])
print("roof", g["roofline"]["frac"], g["roofline"]["call_ms"], [(r["method"], r["frac"], r["call_ms"]) for r in g["roofline_by_method"]])
print("seam", d["seam"]["after"], d["seam"]["speedup"]); print("cpu", d["cpu_baseline"]["value"], d["cpu_baseline"]["seconds"], g["cpu_baseline"]["value"])
PY
