#!/usr/bin/env python3
"""One-off: the MSE search of two builds of the library on the same seeded matrices, bit for bit.
usage: lab_mse_compare.py <lib.so> <out.pt> [<other.pt> to compare with]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from onnx_quantize_amd.hip import _lib  # noqa: E402

_lib.LIB_PATH = os.path.abspath(sys.argv[1])
from onnx_quantize_amd.hip import ops  # noqa: E402

torch.cuda.set_device(0)
out = {}
gen = torch.Generator(device="cuda").manual_seed(5)
for name, shape, dist in (("normal", (4096, 11008), "n"), ("heavy", (2048, 4096), "t")):
    w = torch.randn(shape, generator=gen, device="cuda")
    if dist == "t":
        w = w / (torch.rand(shape, generator=gen, device="cuda") + 0.05)
    for qt, sym in (("uint4", False), ("int4", True), ("uint8", False)):
        for g in (128, 64, 32):
            q, s, z = ops.rtn_quantize(w, qt, "group", g, sym, mse=True)
            out[f"{name}_{qt}_{g}"] = (q.cpu(), s.cpu(), z.cpu())
torch.cuda.synchronize()
if len(sys.argv) > 3:
    prev = torch.load(sys.argv[3])
    bad = [k for k in out if not all(torch.equal(a, b) for a, b in zip(out[k], prev[k]))]
    print("cases", len(out), "different", bad)
    sys.exit(1 if bad else 0)
torch.save(out, sys.argv[2])
print("saved", len(out))
