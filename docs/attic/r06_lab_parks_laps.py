#!/usr/bin/env python3
"""Lab: where rtn_resident_parks spends its time.  Needs a library built with -DOQ_TENSOR_STAMPS: every workgroup sums, over its
iterations, the 100 MHz wall clock spent in each phase of the loop.
usage: lab_parks_laps.py <lib.so> [KxN]"""
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
tiles = st[:, 7]
names = {0: "A + wait for V + local fold", 1: "pending add, ticket, barrier 1", 2: "key atomics, move, issue, barrier 2",
         3: "(blocking branch,) key loads issued", 4: "keys landed, parameters", 6: "K1 + stores", 5: "whole kernel"}
print(f"{k}x{n}: {len(st)} workgroups, tiles per workgroup min/mean/max {tiles.min()}/{tiles.mean():.2f}/{tiles.max()}")
for i, nm in names.items():
    c = st[:, i] / 100.0
    print(f"  {nm:38s} min {c.min():7.2f}  median {np.median(c):7.2f}  max {c.max():7.2f} us   per tile {np.median(c) / max(tiles.mean(), 1):6.2f}")
