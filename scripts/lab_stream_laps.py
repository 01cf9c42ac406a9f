#!/usr/bin/env python3
"""Lab: where the streamed channel kernel spends its time.  Needs a library built with -DOQ_TENSOR_STAMPS: every workgroup sums,
over its tiles, the 100 MHz wall clock spent in: prologue (two tiles loaded + published), wait for the range, keys + ticket,
row loop (stores + refill loads issued), fold + publish of the refilled tile.
usage: lab_stream_laps.py <lib.so> [KxN]"""
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
lib = C.CDLL(_lib.LIB_PATH)
buf = (C.c_uint64 * (512 * 8))()
assert lib.oq_lab_tensor_stamps(buf) == 0
st = np.frombuffer(buf, dtype=np.uint64).reshape(512, 8).astype(np.int64)
st = st[st[:, 5] > 0]
us = st[:, :6] / 100.0
names = ["(unused)", "wait", "keys+ticket", "row loops", "fold+publish", "whole kernel"]
print(f"{k}x{n}: {len(st)} workgroups, refills per workgroup min/mean/max {st[:, 6].min()}/{st[:, 6].mean():.2f}/{st[:, 6].max()}")
for i, nm in enumerate(names):
    c = us[:, i]
    print(f"  {nm:13s} min {c.min():7.2f}  median {np.median(c):7.2f}  max {c.max():7.2f} us   per call of finish {np.median(c) / max(st[:, 6].mean(), 1):6.2f}")
