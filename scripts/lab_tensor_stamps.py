#!/usr/bin/env python3
"""Lab: where the per-tensor one-pass kernel spends its time.  Needs a library built with -DOQ_TENSOR_STAMPS (every workgroup
stamps the 100 MHz wall clock at: start, end of phase A, `go` seen, kept tiles stored, bitmap ready, end).
usage: lab_tensor_stamps.py <lib.so> [KxN]"""
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
    out = ops.rtn_quantize(ws[i % 3], "int8", "tensor")
torch.cuda.synchronize()
lib = C.CDLL(_lib.LIB_PATH)
buf = (C.c_uint64 * (512 * 8))()
assert lib.oq_lab_tensor_stamps(buf) == 0
st = np.frombuffer(buf, dtype=np.uint64).reshape(512, 8).astype(np.int64)
live = st[:, 0] > 0
st = st[live]
t0 = st[:, 0].min()
us = (st[:, :6] - t0) / 100.0
if st[:, 7].max() > 0:
    print(f"  previous call's last end -> this call's first start: {(t0 - st[:, 7].max()) / 100.0:.2f} us (first end {(t0 - st[:, 7].min()) / 100.0:.2f} us before)")
names = ["start", "phase A done", "go seen", "kept stored", "parked+bitmap", "end"]
print(f"{k}x{n}: {live.sum()} workgroups, tiles per workgroup min/mean/max {st[:, 6].min()}/{st[:, 6].mean():.2f}/{st[:, 6].max()}")
for i, nm in enumerate(names):
    c = us[:, i]
    print(f"  {nm:14s} min {c.min():7.2f}  p10 {np.percentile(c, 10):7.2f}  median {np.median(c):7.2f}  p90 {np.percentile(c, 90):7.2f}  max {c.max():7.2f} us")
