#!/usr/bin/env python3
"""Two builds of the library on the same seeded matrices, bit for bit: the MSE range search and the HQQ zero-point walk
(used when their register kernels changed arithmetic form: division-free levels, refined-reciprocal quotient).
usage: lab_build_compare.py <lib.so> <out.pt> [<other.pt> to compare with]     -- also prints ms per call on 4096x11008"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from onnx_quantize_amd.hip import _lib  # noqa: E402

_lib.LIB_PATH = os.path.abspath(sys.argv[1])
from onnx_quantize_amd.hip import ops  # noqa: E402

torch.cuda.set_device(0)


def timed(fn, reps=5):
    fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


out = {}
gen = torch.Generator(device="cuda").manual_seed(5)
for name, shape, dist in (("normal", (4096, 11008), "n"), ("heavy", (2048, 4096), "t")):
    w = torch.randn(shape, generator=gen, device="cuda")
    if dist == "t":
        w = w / (torch.rand(shape, generator=gen, device="cuda") + 0.05)
    for qt, sym in (("uint4", False), ("int4", True), ("uint8", False)):
        for g in (128, 64, 32):
            q, s, z = ops.rtn_quantize(w, qt, "group", g, sym, mse=True)
            out[f"mse_{name}_{qt}_{g}"] = (q.cpu(), s.cpu(), z.cpu())
    for g in (128, 64, 32, 16, 256):
        for kw in ({}, {"early_stop": False, "iters": 7}, {"per_round_launches": True}):
            r = ops.hqq_quantize(w, g, **kw)
            out[f"hqq_{name}_{g}_{sorted(kw)}"] = tuple(t.cpu() for t in r[:3]) + (torch.tensor(int(r[3])),)
    if name == "normal":
        for g in (128, 64, 32):
            print(f"g={g}: mse {timed(lambda: ops.rtn_quantize(w, 'uint4', 'group', g, mse=True)):.3f} ms, "
                  f"hqq {timed(lambda: ops.hqq_quantize(w, g)):.3f} ms", flush=True)
torch.cuda.synchronize()
if len(sys.argv) > 3:
    prev = torch.load(sys.argv[3])
    bad = [k for k in out if not all(torch.equal(a, b) for a, b in zip(out[k], prev[k]))]
    print("cases", len(out), "different", bad)
    sys.exit(1 if bad else 0)
torch.save(out, sys.argv[2])
print("saved", len(out))
