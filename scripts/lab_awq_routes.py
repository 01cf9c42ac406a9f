#!/usr/bin/env python3
"""Lab: direct vs Gram route of the AWQ searches over the number of calibration rows (two builds: the second with
-DOQ_AWQ_GRAM_RATIO=1000 never takes the Gram route).  usage: lab_awq_routes.py <lib.so>"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from onnx_quantize_amd.hip import _lib  # noqa: E402

_lib.LIB_PATH = os.path.abspath(sys.argv[1])
from onnx_quantize_amd.hip import ops  # noqa: E402

torch.cuda.set_device(0)
k = n = 4096
gen = torch.Generator(device="cuda").manual_seed(3)
w = torch.randn((k, n), generator=gen, device="cuda") * 0.02
for t in (4096, 6144, 8192, 12288, 16384, 24576):
    x = torch.randn((t, k), generator=gen, device="cuda") * (0.1 + 3.9 * torch.rand(k, generator=gen, device="cuda"))
    res = []
    for fn in (lambda: ops.awq_scale_search(x, w, "uint4", "group", 128), lambda: ops.awq_clip_search(x, w, "uint4", "group", 128)):
        r = fn(); torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(3):
            fn()
        b.record(); torch.cuda.synchronize()
        res.append((a.elapsed_time(b) / 3, r))
    print(f"T={t}: scale {res[0][0]:.2f} ms (best loss {float(min(res[0][1][1])):.8f}), clip {res[1][0]:.2f} ms (ratio {res[1][1][0]})", flush=True)
    del x
