"""K1 / K2 elementwise entry points (oq_quantize_f32 / oq_dequantize_f32) on the headline matrix.  python scripts/quick_elementwise.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from onnx_quantize_amd.hip import ops

k, n, g = 4096, 11008, 128
w = torch.randn((k, n), device="cuda")


def timeit(fn, reps=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


for qtype, strategy, mode, group in (("uint4", "group", "group", g), ("int8", "channel", "col", 1), ("int8", "tensor", "tensor", 1)):
    q, s, z = ops.rtn_quantize(w, qtype, strategy, g if strategy == "group" else -1)
    tq = timeit(lambda: ops.quantize(w, s, z, qtype, False, False, mode, group))
    td = timeit(lambda: ops.dequantize(q, s, z, qtype, mode, group))
    assert torch.equal(ops.quantize(w, s, z, qtype, False, False, mode, group), q)
    moved_q, moved_d = k * n * 5, k * n * 5
    print(f"{qtype:6s} {strategy:8s} quantize {tq:8.1f} us ({moved_q / tq / 1e6:5.2f} TB/s)   dequantize {td:8.1f} us ({moved_d / td / 1e6:5.2f} TB/s)")
