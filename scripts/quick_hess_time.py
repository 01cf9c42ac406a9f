import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from onnx_quantize_amd.hip import ops
for K, T in ((4096, 16384), (4096, 65536), (11008, 8192), (11008, 65536)):
    x = torch.randn((8, T // 8, K), device="cuda")
    h = torch.zeros((K, K), device="cuda")
    for _ in range(2): ops.hessian_accumulate(x, h, 0)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(3): ops.hessian_accumulate(x, h, 8)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 3
    print(f"hessian K={K} T={T}: {ms:.2f} ms  {T*K*K/ms/1e9:.1f} TFLOP/s executed", flush=True)
    del x, h
