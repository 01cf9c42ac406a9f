"""Merged launch vs separate launches at model footprint, per shape (outputs distinct, > Infinity Cache)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from onnx_quantize_amd.hip import ops
def t(fn, reps=3):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
for k, n, cnt in ((4096, 4096, 128), (4096, 11008, 64), (11008, 4096, 32), (2048, 2048, 256), (8192, 8192, 32)):
    big = torch.randn((cnt, k, n), device="cuda")
    q = torch.empty((cnt, n, k // 128, 64), dtype=torch.uint8, device="cuda"); s = torch.empty((cnt, n * k // 128, 1), device="cuda"); z = torch.empty((cnt, n * k // 128, 1), dtype=torch.uint8, device="cuda")
    alg = cnt * k * n * (4 + 0.5 + 5 / 128)
    for m in (cnt, 16, 8, 4, 2, 1):
        def run():
            for i in range(0, cnt, m):
                ops.rtn_quantize_batched(big[i:i + m], "uint4", 128, layout="nbits", out=(q[i:i + m], s[i:i + m], z[i:i + m]))
        ms = t(run)
        print(f"{k}x{n} x{cnt}, {m} per launch: {ms / cnt * 1e3:.2f} us per matrix, {alg / ms / 1e9 / 8000 * 1e3:.3f} of peak", flush=True)
    del big, q, s, z
