#!/usr/bin/env python3
"""int4 g128 RTN of the headline matrix in the three output layouts ([K,N] bytes, [K,N/2] nibble pairs, MatMulNBits blob for
uint4): us per call over rotating inputs, fraction of 8 TB/s in algorithmic bytes (W once + N K / 2 + 5 B per group)."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from onnx_quantize_amd.hip import ops  # noqa: E402

torch.cuda.set_device(0)
k, n, g = 4096, 11008, 128
ws = [torch.randn((k, n), device="cuda") for _ in range(4)]
alg = k * n * 4 + k * n // 2 + (k * n // g) * 5
for qtype, layout in (("int4", "kn"), ("int4", "kn_packed4"), ("uint4", "kn_packed4"), ("uint4", "nbits")):
    outs = ops.rtn_quantize(ws[0], qtype, "group", g, layout=layout)
    for i in range(10):
        ops.rtn_quantize(ws[i % 4], qtype, "group", g, layout=layout, out=outs)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for i in range(200):
        ops.rtn_quantize(ws[i % 4], qtype, "group", g, layout=layout, out=outs)
    b.record()
    torch.cuda.synchronize()
    us = a.elapsed_time(b) * 1e3 / 200
    print(json.dumps(dict(qtype=qtype, layout=layout, us=round(us, 2), frac=round(alg / us / 1e6 / 8.0, 4))), flush=True)
