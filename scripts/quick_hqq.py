import os, sys, time, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from onnx_quantize_amd.hip import ops
k, n, g = 4096, 11008, 128
w = torch.randn((k, n), device="cuda")
for early in (True, False):
    ops.hqq_quantize(w, g, early_stop=early)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(3):
        q, s, z, rounds = ops.hqq_quantize(w, g, early_stop=early)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 3
    print(f"hqq uint4 g{g} {k}x{n} early_stop={early}: {ms:.2f} ms, rounds={int(rounds.item())}, {ms/max(int(rounds.item()),1)*1e3:.0f} us/round, {k*n/ms/1e3:.0f} M-param/s", flush=True)
