"""Steady state of the seam with and without prefetch / page locking (run with PYTHONPATH=.)."""
import os
import sys
import time
import numpy as np
import torch

from onnx_quantize_amd import QConfig, QuantType, QWeightArgs, seam
from onnx_quantize_amd.staging import default_stager

base = np.random.default_rng(0).standard_normal((4096, 11008), dtype=np.float32)
qc = QConfig(weights=QWeightArgs(dtype=QuantType.QUInt4, group_size=128))


class T:
    def __init__(self, a): self._a = a
    def numpy(self): return self._a


class V:
    def __init__(self, n, a): self.name, self.const_value = n, T(a)


st = default_stager()
print("lock_pages", st.lock_pages)
K = 8
held = []
for rep in range(4):
    ws = [base.copy() for _ in range(K)]
    t0 = time.perf_counter()
    per = []
    for i, w in enumerate(ws):
        t1 = time.perf_counter()
        held.append(seam.weight_arrays(V(f"w{i}", w), qc, None, True))
        per.append(round((time.perf_counter() - t1) * 1e3, 1))
    print("on demand  ms/weight", round((time.perf_counter() - t0) * 1e3 / K, 2), per)
for rep in range(4):
    ws = [base.copy() for _ in range(K)]
    t0 = time.perf_counter()
    st.prefetch([(f"w{i}", w) for i, w in enumerate(ws)])
    per = []
    for i, w in enumerate(ws):
        t1 = time.perf_counter()
        held.append(seam.weight_arrays(V(f"w{i}", w), qc, None, True))
        per.append(round((time.perf_counter() - t1) * 1e3, 1))
    print("prefetched ms/weight", round((time.perf_counter() - t0) * 1e3 / K, 2), per)
