import sys, time, numpy as np, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from onnx_quantize_amd.hip import ops, _lib as L
import ctypes as C
k, n, g = 4096, 11008, 128
ws = [torch.randn((k, n), device="cuda") for _ in range(4)]
for layout in ("kn", "nbits"):
    outs = [ops.rtn_quantize(w, "uint4", "group", g, layout=layout) for w in ws]
    torch.cuda.synchronize()
    for rot in (1, 4):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        iters = 200
        for i in range(20):
            ops.rtn_quantize(ws[i % rot], "uint4", "group", g, layout=layout, out=outs[i % rot])
        e0.record()
        for i in range(iters):
            ops.rtn_quantize(ws[i % rot], "uint4", "group", g, layout=layout, out=outs[i % rot])
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / iters * 1e3
        print(f"layout={layout} rotate={rot}: {us:.1f} us/launch  {204660736/us/1e6:.2f} TB/s algorithmic  {k*n/us/1e6:.3f} T-param/s")
