#!/usr/bin/env python3
"""Lab: the AWQ / clip searches' losses with the one-product (first fp16 pieces only) and the three-product (22-bit) GEMM.
usage: lab_awq_loss_forms.py <lib.so> <out.pt> [<other.pt>]   -- run once per build (the second with -DOQ_AWQ_HI_ONLY=0)"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from onnx_quantize_amd.hip import _lib  # noqa: E402

_lib.LIB_PATH = os.path.abspath(sys.argv[1])
from onnx_quantize_amd.hip import ops  # noqa: E402

torch.cuda.set_device(0)
gen = torch.Generator(device="cuda").manual_seed(9)
out = {}
for name, (t, k, n), heavy in (("llama", (4096, 4096, 4096), False), ("outliers", (2048, 4096, 11008), True), ("small", (300, 512, 768), True)):
    x = torch.randn((t, k), generator=gen, device="cuda") * (0.1 + 3.9 * torch.rand(k, generator=gen, device="cuda"))
    if heavy:
        x[:, ::97] *= 60.0                       # outlier channels, what AWQ exists for
    w = torch.randn((k, n), generator=gen, device="cuda") * 0.02
    for qt, strat, g in (("uint4", "group", 128), ("int8", "channel", -1)):
        s, losses = ops.awq_scale_search(x, w, qt, strat, g)
        r, closses = ops.awq_clip_search(x, w, qt, strat, g)
        out[f"{name}_{qt}"] = (torch.as_tensor(losses).double().cpu(), torch.as_tensor(closses).double().cpu(), torch.as_tensor(s).cpu(), torch.tensor(float(r)))
    if name == "llama":
        for fn, nm in ((lambda: ops.awq_scale_search(x, w, "uint4", "group", 128), "scale"), (lambda: ops.awq_clip_search(x, w, "uint4", "group", 128), "clip")):
            fn(); torch.cuda.synchronize()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(3):
                fn()
            b.record(); torch.cuda.synchronize()
            print(f"{nm} search 4096^3: {a.elapsed_time(b) / 3:.2f} ms", flush=True)
if len(sys.argv) > 3:
    prev = torch.load(sys.argv[3])
    for kx in out:
        l0, c0, s0, r0 = prev[kx]
        l1, c1, s1, r1 = out[kx]
        print(f"{kx}: scale losses max rel diff {float(((l1 - l0).abs() / l0).max()):.2e}, clip losses {float(((c1 - c0).abs() / c0).max()):.2e}, "
              f"same grid point {int(l0.argmin()) == int(l1.argmin())}, same scales {bool(torch.equal(s0, s1))}, same clip ratio {float(r0) == float(r1)}; "
              f"gap of the two best scale losses {float((l0.sort().values[1] - l0.min()) / l0.min()):.2e}")
else:
    torch.save(out, sys.argv[2])
    print("saved")
