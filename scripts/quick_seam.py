"""The NumPy seam (`_rtn_quantize(array, ...)`, rtn.py:54-65 signature) end to end on the headline matrix: host array in, host arrays out."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from onnx_quantize_amd.algorithms.rtn import _rtn_quantize
from onnx_quantize_amd.dtypes import QuantType
from onnx_quantize_amd.config import QuantizationStrategy

w = np.random.default_rng(0).standard_normal((4096, 11008), dtype=np.float32)
for _ in range(2):
    _rtn_quantize(w, QuantType.QUInt4, QuantizationStrategy.GROUP, 128, False, False, 1.0, False, np.float32, np.uint8)
ts = []
for _ in range(5):
    t0 = time.perf_counter()
    q, s, z = _rtn_quantize(w, QuantType.QUInt4, QuantizationStrategy.GROUP, 128, False, False, 1.0, False, np.float32, np.uint8)
    ts.append(time.perf_counter() - t0)
print(f"_rtn_quantize host->host 4096x11008 uint4 g128: best {min(ts) * 1e3:.2f} ms, median {sorted(ts)[2] * 1e3:.2f} ms; out {q.shape} {q.dtype}")
x = torch.from_numpy(w)
t0 = time.perf_counter(); xd = x.cuda(); torch.cuda.synchronize(); t1 = time.perf_counter()
print(f"pageable H2D 180 MB: {(t1 - t0) * 1e3:.2f} ms")
xp = x.pin_memory()
t0 = time.perf_counter(); xd = xp.cuda(non_blocking=True); torch.cuda.synchronize(); t1 = time.perf_counter()
print(f"pinned H2D 180 MB: {(t1 - t0) * 1e3:.2f} ms")
qd = torch.empty((4096, 11008), dtype=torch.uint8, device="cuda")
t0 = time.perf_counter(); qh = qd.cpu(); t1 = time.perf_counter()
print(f"pageable D2H 45 MB: {(t1 - t0) * 1e3:.2f} ms")
