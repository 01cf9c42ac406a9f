"""Footprint vs duration: strided batch of 4096x11008 uint4 g128 blob."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from onnx_quantize_amd.hip import ops
def t(fn, reps):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
k, n = 4096, 11008
big = torch.randn((64, k, n), device="cuda")
out = None
for cnt, reps in ((4, 40), (8, 5), (8, 40), (16, 20), (32, 10), (64, 5)):
    stack = big[:cnt]
    q = torch.empty((cnt, n, k // 128, 64), dtype=torch.uint8, device="cuda"); s = torch.empty((cnt, n * k // 128, 1), device="cuda"); z = torch.empty((cnt, n * k // 128, 1), dtype=torch.uint8, device="cuda")
    ms = t(lambda: ops.rtn_quantize_batched(stack, "uint4", 128, layout="nbits", out=(q, s, z)), reps)
    alg = cnt * k * n * (4 + 0.5 + 5 / 128)
    print(f"x{cnt} reps {reps}: {ms / cnt * 1e3:.2f} us per matrix, {alg / ms / 1e9 / 8000 * 1e3:.3f} of peak", flush=True)
# the same 64 matrices, but 8 at a time in sequence (footprint 13 GB, launches of 8)
def seq8():
    for i in range(0, 64, 8):
        ops.rtn_quantize_batched(big[i:i + 8], "uint4", 128, layout="nbits", out=(q[i:i + 8], s[i:i + 8], z[i:i + 8]))
ms = t(seq8, 5)
print(f"64 matrices as 8 launches of 8: {ms / 64 * 1e3:.2f} us per matrix, {64 * k * n * (4 + 0.5 + 5 / 128) / ms / 1e9 / 8000 * 1e3:.3f} of peak")
