import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from onnx_quantize_amd.hip import ops
k, n = 4096, 4096
for t in (4096, 32768):        # 32768 rows (16 sequences of 2048 tokens): the Gram route (T >= 6 K)
    x = torch.randn((t, k), device="cuda") * (0.1 + 3.9 * torch.rand(k, device="cuda"))
    w = torch.randn((k, n), device="cuda") * 0.02
    for name, fn in (("awq_scale_search (20 points)", lambda: ops.awq_scale_search(x, w, "uint4", "group", 128)),
                     ("awq_clip_search (10 points)", lambda: ops.awq_clip_search(x, w, "uint4", "group", 128)),
                     ("smooth_quant_scale", lambda: ops.smooth_quant_scale(x, w, 0.5))):
        fn(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3): fn()
        torch.cuda.synchronize()
        print(f"{name} K={k} N={n} T={t}: {(time.perf_counter() - t0) / 3 * 1e3:.2f} ms", flush=True)
    del x
