set -u
O=gpurun_out/r04_d; mkdir -p $O
timeout -k 5 300 python scripts/quick_searches.py > $O/searches.log 2>&1 || { echo SEARCHES FAILED; tail -20 $O/searches.log; exit 1; }
grep -v amdgpu $O/searches.log
timeout -k 10 600 python -m pytest tests/test_hqq.py tests/test_mse_gpu.py tests/test_rtn_gpu.py -m gpu -q -x > $O/pytest.log 2>&1 || { echo PYTEST FAILED; tail -30 $O/pytest.log; exit 1; }
tail -2 $O/pytest.log
timeout -k 10 300 python - <<'PY' > $O/model_rtn.log 2>&1
import json, sys, torch
sys.path.insert(0, '.')
import bench
dev = torch.device('cuda', 0)
w = torch.randn((4096, 11008), device=dev)
from onnx_quantize_amd.hip import ops
out = ops.rtn_quantize(w, "uint4", "group", 128, layout="nbits")
r = bench.model_rtn_bench(dev, "nbits", out, w)
print(json.dumps({k: r[k] for k in ("device_ms", "host_wall_ms", "frac", "per_matrix_loop_frac", "equals_single_matrix_outputs")}))
print(json.dumps(r["small_matrices"]))
PY
grep -v amdgpu $O/model_rtn.log | tail -3
