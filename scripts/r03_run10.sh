#!/bin/bash
set -u
cd $GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out/r03_run10
mkdir -p $OUT
timeout -k 10 900 python bench.py --steps 20 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"; tail -c 300 $OUT/bench.err
python - <<'PY'
import json, os
d = json.load(open(os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/r03_run10/bench.json"))
g = d["gptq"]
print("headline", d["value"], d["roofline"]["frac"], "other", d["other_layout"]["frac"], "batched", d["batched_launch"]["frac"])
print("model_rtn", d["model_rtn"]["frac"], d["model_rtn"]["device_ms"], d["model_rtn"]["per_matrix_loop_frac"], d["model_rtn"]["small_matrices"]["speedup"])
print("calib", d["calibration"]["value"], d["calibration"]["roofline"]["frac"], d["calibration"]["verified"])
print("gptq", g["value"], g["seconds"], g["verified"])
print("corrected", g["corrected"]["seconds"], g["corrected"]["verified"], [s["ratio"] for s in g["corrected"]["verification"]])
print("by method", g["wall_by_hessian_method"])
print("seam", d["seam"]["after"]["ms_per_weight_trials"], d["seam"]["speedup"])
PY
