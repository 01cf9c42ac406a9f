"""Phase timing of the device-resident seam on fresh 4096x11008 weights."""
import time
import numpy as np
import torch

from onnx_quantize_amd import QConfig, QuantType, QWeightArgs, seam
from onnx_quantize_amd.hip import ops
from onnx_quantize_amd.staging import default_stager, _identity

base = np.random.default_rng(0).standard_normal((4096, 11008), dtype=np.float32)
qc = QConfig(weights=QWeightArgs(dtype=QuantType.QUInt4, group_size=128))


class T:
    def __init__(self, a): self._a = a
    def numpy(self): return self._a


class V:
    def __init__(self, n, a): self.name, self.const_value = n, T(a)


st = default_stager()
seam.weight_arrays(V("warm", base.copy()), qc, None, True)
torch.cuda.synchronize()
for rep in range(2):
    ws = [base.copy() for _ in range(6)]
    tt = {"identity": 0, "upload": 0, "kernel": 0, "packzp": 0, "dl_blob": 0, "dl_scale": 0, "dl_zp": 0}
    t_all = time.perf_counter()
    for i, w in enumerate(ws):
        t0 = time.perf_counter(); _identity(w); tt["identity"] += time.perf_counter() - t0
        t0 = time.perf_counter(); wd = st.upload(w); torch.cuda.synchronize(); tt["upload"] += time.perf_counter() - t0
        t0 = time.perf_counter(); q, s, z = ops.rtn_quantize(wd, "uint4", "group", 128, layout="nbits"); torch.cuda.synchronize(); tt["kernel"] += time.perf_counter() - t0
        t0 = time.perf_counter(); pz = ops.pack_zero_points_u4(z.reshape(-1), 11008, 32); torch.cuda.synchronize(); tt["packzp"] += time.perf_counter() - t0
        t0 = time.perf_counter(); b = st.download(q); tt["dl_blob"] += time.perf_counter() - t0
        t0 = time.perf_counter(); sc = st.download(s); tt["dl_scale"] += time.perf_counter() - t0
        t0 = time.perf_counter(); zz = st.download(pz); tt["dl_zp"] += time.perf_counter() - t0
    print("phases ms/weight", {k: round(v * 1e3 / 6, 2) for k, v in tt.items()}, "total", round((time.perf_counter() - t_all) * 1e3 / 6, 2))
    ws = [base.copy() for _ in range(6)]
    t0 = time.perf_counter()
    keep = [seam.weight_arrays(V(f"w{i}", w), qc, None, True) for i, w in enumerate(ws)]
    print("weight_arrays on demand ms/weight", round((time.perf_counter() - t0) * 1e3 / 6, 2))
    ws = [base.copy() for _ in range(6)]
    t0 = time.perf_counter()
    st.prefetch([(f"w{i}", w) for i, w in enumerate(ws)])
    keep = [seam.weight_arrays(V(f"w{i}", w), qc, None, True) for i, w in enumerate(ws)]
    print("weight_arrays prefetched ms/weight", round((time.perf_counter() - t0) * 1e3 / 6, 2), st.stats)
