import os, sys, time, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from onnx_quantize_amd.hip import ops
k, n, g = 4096, 11008, 128
ws = [torch.randn((k, n), device="cuda") for _ in range(4)]
def timeit(label, fn, alg=204660736):
    for rot in (1, 4):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        iters = 200
        for i in range(20): fn(i % rot)
        e0.record()
        for i in range(iters): fn(i % rot)
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / iters * 1e3
        print(f"{label} rotate={rot}: {us:.1f} us/launch  {alg/us/1e6:.2f} TB/s algorithmic  {k*n/us/1e6:.3f} T-param/s", flush=True)
for layout in ("kn", "nbits"):
    outs = [ops.rtn_quantize(w, "uint4", "group", g, layout=layout) for w in ws]
    timeit(f"uint4 g128 layout={layout}", lambda j: ops.rtn_quantize(ws[j], "uint4", "group", g, layout=layout, out=outs[j]))
outs = [ops.rtn_quantize(w, "uint4", "group", g, emit_q=False) for w in ws]
timeit("qparams-only (read ceiling of this access pattern)", lambda j: ops.rtn_quantize(ws[j], "uint4", "group", g, emit_q=False, out=outs[j]), alg=k*n*4)
outs = [ops.rtn_quantize(w, "uint4", "group", g, symmetric=True) for w in ws]
timeit("uint4 g128 sym kn", lambda j: ops.rtn_quantize(ws[j], "uint4", "group", g, symmetric=True, out=outs[j]))
outs = [ops.rtn_quantize(w, "int8", "group", g) for w in ws]
timeit("int8 g128 kn", lambda j: ops.rtn_quantize(ws[j], "int8", "group", g, out=outs[j]))
for g2 in (32, 64, 256):
    outs = [ops.rtn_quantize(w, "uint4", "group", g2) for w in ws]
    timeit(f"uint4 g{g2} kn", lambda j: ops.rtn_quantize(ws[j], "uint4", "group", g2, out=outs[j]))
outs = [ops.rtn_quantize(w, "int8", "channel") for w in ws]
timeit("int8 channel (two-pass)", lambda j: ops.rtn_quantize(ws[j], "int8", "channel", out=outs[j]))
outs = [ops.rtn_quantize(w, "int8", "tensor") for w in ws]
timeit("int8 tensor (two-pass)", lambda j: ops.rtn_quantize(ws[j], "int8", "tensor", out=outs[j]))
# copy ceiling
dst = [torch.empty_like(w) for w in ws]
timeit("torch copy_ fp32 (read+write 360 MB)", lambda j: dst[j].copy_(ws[j]), alg=2*k*n*4)
