"""One line per device entry point on the headline matrix (4096 x 11008 fp32): time and the bytes it has to move.
python scripts/quick_all_entries.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from onnx_quantize_amd.hip import ops

k, n, g = 4096, 11008, 128
w = torch.randn((k, n), device="cuda")
mb = k * n / 1e6


def timeit(fn, reps=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


def line(name, us, mbytes):
    print(f"{name:58s} {us:9.1f} us   {mbytes / us:6.2f} TB/s of {mbytes:7.1f} MB", flush=True)


for qtype in ("uint4", "int8"):
    for strategy, gs in (("group", g), ("channel", -1), ("tensor", -1)):
        for layout in (("kn", "nbits") if strategy == "group" else ("kn",)):
            us = timeit(lambda: ops.rtn_quantize(w, qtype, strategy, gs, layout=layout))
            out = mb * (0.5 if layout == "nbits" and qtype == "uint4" else 1.0)
            passes = 1 if strategy == "group" else 2
            line(f"rtn_quantize {qtype} {strategy} {layout}", us, mb * 4 * passes + out)
        us = timeit(lambda: ops.rtn_quantize(w, qtype, strategy, gs, emit_q=False))
        line(f"rtn qparams only {qtype} {strategy}", us, mb * 4)
q, s, z = ops.rtn_quantize(w, "uint4", "group", g)
line("pack_nibbles (45 M values)", timeit(lambda: ops.pack_nibbles(q)), mb * 1.5)
line("absmax columns", timeit(lambda: ops.absmax(w)), mb * 4)
line("absmax rows", timeit(lambda: ops.absmax(w, per_row=True)), mb * 4)
st = ops.minmax_state(w.device)
line("minmax_collect (one tensor)", timeit(lambda: ops.minmax_collect(w, st)), mb * 4)
line("rtn mse=True uint4 group", timeit(lambda: ops.rtn_quantize(w, "uint4", "group", g, mse=True), reps=3), mb * 4.5)
line("hqq uint4 group 20 rounds", timeit(lambda: ops.hqq_quantize(w, g), reps=3), mb * 4.5)
