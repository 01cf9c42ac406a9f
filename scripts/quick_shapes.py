"""RTN uint4 g128 (MatMulNBits blob) on the MatMul shapes of Llama-2-7B / gemma-3-270m: time and algorithmic TB/s."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from onnx_quantize_amd.hip import ops
shapes = [(4096, 4096), (4096, 11008), (11008, 4096), (4096, 32000), (640, 2048), (2048, 640), (640, 1024), (8192, 28672)]
for k, n in shapes:
    nb = max(2, int(600e6 // (k * n * 4)) + 1)          # rotate over > 256 MiB of inputs
    ws = [torch.randn((k, n), device="cuda") for _ in range(min(nb, 8))]
    outs = [ops.rtn_quantize(w, "uint4", "group", 128, layout="nbits") for w in ws]
    for layout in ("nbits", "kn"):
        outs = [ops.rtn_quantize(w, "uint4", "group", 128, layout=layout) for w in ws]
        torch.cuda.synchronize()
        iters = 100
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(iters):
            ops.rtn_quantize(ws[i % len(ws)], "uint4", "group", 128, layout=layout, out=outs[i % len(ws)])
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / iters * 1e3
        alg = k * n * (4 + 0.5) + (k // 128) * n * 5
        print(f"{k:6d} x {n:6d} {layout:5s}: {us:8.1f} us  {alg / us / 1e6:5.2f} TB/s algorithmic  ({alg / us / 1e6 / 8 * 100:4.1f} % of 8 TB/s)", flush=True)
    del ws, outs
