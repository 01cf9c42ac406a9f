#!/usr/bin/env python3
"""The model-scale RTN call alone (bench.model_rtn_bench's workload: Llama-2-7B's 224 weights, uint4 g128 blob), three times."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from onnx_quantize_amd.hip import ops  # noqa: E402

dev = torch.device("cuda", 0)
gen = torch.Generator(device=dev).manual_seed(5)
shapes = [(4096, 4096)] * 4 + [(4096, 11008)] * 2 + [(11008, 4096)]
base = {sh: torch.randn(sh, generator=gen, device=dev) * 0.02 for sh in set(shapes)}
ws = [base[sh].clone() for _ in range(32) for sh in shapes]
for rep in range(3):
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    e0.record()
    res = ops.rtn_quantize_many(ws, "uint4", 128, layout="nbits")
    t1 = time.perf_counter()
    e1.record()
    torch.cuda.synchronize()
    print(f"rep {rep}: device {e0.elapsed_time(e1):.3f} ms, host issue {1e3 * (t1 - t0):.3f} ms", flush=True)
    del res
