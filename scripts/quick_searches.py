#!/usr/bin/env python3
"""MSE range search and HQQ zero-point optimisation on the headline matrix (uint4, g = 128 / 64): ms per call, and the one-pass
HQQ route against the per-round one (same bits expected)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from onnx_quantize_amd.hip import ops  # noqa: E402

torch.cuda.set_device(0)
w = torch.randn((4096, 11008), device="cuda")


def timed(fn, reps=5):
    fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        r = fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps, r


for g in (128, 64, 32):
    ms, _ = timed(lambda: ops.rtn_quantize(w, "uint4", "group", g, mse=True))
    print(f"mse g={g}: {ms:.3f} ms", flush=True)
    m1, r1 = timed(lambda: ops.hqq_quantize(w, g))
    m2, r2 = timed(lambda: ops.hqq_quantize(w, g, per_round_launches=True))
    same = all(torch.equal(a, b) for a, b in zip(r1[:3], r2[:3])) and int(r1[3]) == int(r2[3])
    print(f"hqq g={g}: one pass {m1:.3f} ms, per round {m2:.3f} ms, rounds {int(r1[3])}, same bits {same}", flush=True)
    m3, r3 = timed(lambda: ops.hqq_quantize(w, g, early_stop=False, iters=7))
    m4, r4 = timed(lambda: ops.hqq_quantize(w, g, early_stop=False, iters=7, per_round_launches=True))
    print(f"   iters=7 no early stop: {m3:.3f} / {m4:.3f} ms, same bits {all(torch.equal(a, b) for a, b in zip(r3[:3], r4[:3]))}", flush=True)
