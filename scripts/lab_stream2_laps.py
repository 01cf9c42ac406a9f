#!/usr/bin/env python3
"""Lab: where rtn_resident_stream2 spends its time.  Needs a library built with -DOQ_TENSOR_STAMPS; per workgroup the kernel sums
the 100 MHz wall clock of wave 0 (communication) and of the first compute wave between their lap points.
usage: OQ_RTN_RES_TILE=2 lab_stream2_laps.py <lib.so> [KxN]"""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from onnx_quantize_amd.hip import _lib  # noqa: E402

_lib.LIB_PATH = os.path.abspath(sys.argv[1])
from onnx_quantize_amd.hip import ops  # noqa: E402

k, n = (int(v) for v in (sys.argv[2] if len(sys.argv) > 2 else "4096x11008").split("x"))
torch.cuda.set_device(0)
ws = [torch.randn((k, n), device="cuda") for _ in range(3)]
for i in range(6):
    out = ops.rtn_quantize(ws[i % 3], "int8", "channel")
torch.cuda.synchronize()
st_, en = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
st_.record()
for i in range(50):
    ops.rtn_quantize(ws[i % 3], "int8", "channel", out=out)
en.record()
torch.cuda.synchronize()
lib = C.CDLL(_lib.LIB_PATH)
buf = (C.c_uint64 * (512 * 8))()
assert lib.oq_lab_tensor_stamps(buf) == 0
st = np.frombuffer(buf, dtype=np.uint64).reshape(512, 8).astype(np.int64)
st = st[st[:, 3] > 0]
names = ["compute: landing + fold", "comm: poll", "comm: keys + ticket", "compute: params + row loop", "comm: publish (+ rare path)",
         "comm: at the barrier", "comm: polls (count)", "compute: at the barrier"]
print(f"{k}x{n}: {st_.elapsed_time(en) * 20:.1f} us per call; {len(st)} workgroups (per-workgroup sums over its tiles, us)")
for i, nm in enumerate(names):
    c = st[:, i] / (1.0 if i == 6 else 100.0)
    print(f"  {nm:30s} min {c.min():7.2f}  median {np.median(c):7.2f}  max {c.max():7.2f}")
