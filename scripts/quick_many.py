"""List RTN (pointer table) against the strided batch and the per-matrix loop, HIP events, uint4 g128 blob."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from onnx_quantize_amd.hip import ops
def t(fn, reps=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
for (k, n, cnt) in ((4096, 11008, 8), (4096, 4096, 32), (11008, 4096, 8), (4096, 11008, 64)):
    stack = torch.randn((cnt, k, n), device="cuda")
    ws = [stack[i] for i in range(cnt)]
    sep = [w.clone() for w in ws]
    alg = cnt * k * n * (4 + 0.5 + 5 / 128)
    for name, fn in (("batched (strides)", lambda: ops.rtn_quantize_batched(stack, "uint4", 128, layout="nbits")),
                     ("ptr table, stacked", lambda: ops.rtn_quantize_many(ws, "uint4", 128, layout="nbits")),
                     ("ptr table, separate", lambda: ops.rtn_quantize_many(sep, "uint4", 128, layout="nbits")),
                     ("per-matrix loop", lambda: [ops.rtn_quantize(w, "uint4", "group", 128, layout="nbits") for w in sep])):
        ms = t(fn)
        print(f"{k}x{n} x{cnt} {name}: {ms:.3f} ms, {alg / ms / 1e9 / 8000 * 1e3:.3f} of peak", flush=True)
    del stack, ws, sep
