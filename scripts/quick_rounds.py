import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from onnx_quantize_amd.hip import ops
K = 4096
import sys as _s
for N in ([int(v) for v in _s.argv[1:]] or (4096, 6144, 8192, 9216, 10240, 11008, 12288, 13312, 14336, 16384)):
    ws = [torch.randn((K, N), device="cuda") for _ in range(max(2, int(1.2e9 / (K * N * 4))))]
    outs = [ops.rtn_quantize(w, "uint4", "group", 128, layout="nbits") for w in ws[:1]]
    out = outs[0]
    for w in ws: ops.rtn_quantize(w, "uint4", "group", 128, layout="nbits", out=out)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 40
    e0.record()
    for i in range(reps): ops.rtn_quantize(ws[i % len(ws)], "uint4", "group", 128, layout="nbits", out=out)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / reps * 1e3
    waves = (K // 128) * (N // 32)
    print(f"N={N:6d} waves={waves:6d} rounds={waves / 4096:5.2f}  {us:7.2f} us   {K * N * 4.539 / us / 1e6:5.2f} TB/s   us/round={us / (waves / 4096):6.2f}", flush=True)
