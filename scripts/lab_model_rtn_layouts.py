#!/usr/bin/env python3
"""Lab: the model-scale RTN call (Llama-2-7B's 224 weights) in the output layouts; OQ_RTN_* knobs apply (speed only)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from onnx_quantize_amd.hip import ops  # noqa: E402

dev = torch.device("cuda", 0)
gen = torch.Generator(device=dev).manual_seed(5)
shapes = [(4096, 4096)] * 4 + [(4096, 11008)] * 2 + [(11008, 4096)]
base = {sh: torch.randn(sh, generator=gen, device=dev) * 0.02 for sh in set(shapes)}
ws = [base[sh].clone() for _ in range(32) for sh in shapes]
params = sum(w.numel() for w in ws)
cases = (("uint4", "nbits", 0.5), ("int4", "kn_packed4", 0.5), ("int8", "kn", 1.0))
if len(sys.argv) > 1:
    cases = tuple(c for c in cases if c[1] in sys.argv[1].split(","))
for qtype, layout, qbytes in cases:
    alg = params * 4 + params * qbytes + params / 128 * 5
    best = 1e9
    for rep in range(4):
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        res = ops.rtn_quantize_many(ws, qtype, 128, layout=layout)
        e1.record()
        torch.cuda.synchronize()
        if rep:
            best = min(best, e0.elapsed_time(e1))
        del res
    print(f"{qtype} {layout}: {best:.3f} ms, {alg / best / 1e9:.2f} TB/s = {alg / best / 1e9 / 8:.3f} of peak", flush=True)
