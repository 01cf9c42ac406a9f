#!/usr/bin/env python3
"""Lab: does a re-read of data that was streamed a moment ago come from the 256 MiB Infinity Cache?  The calibration min-max kernel
(a pure read at 5.9 TB/s from HBM) reads a 180 MB tensor, then parts of it again."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from onnx_quantize_amd.hip import ops  # noqa: E402

torch.cuda.set_device(0)
big = torch.randn((4096, 11008), device="cuda")
other = [torch.randn((4096, 11008), device="cuda") for _ in range(3)]


def timed(fn, reps=1):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3 / reps


state = ops.minmax_state(big.device)


def rd(t):
    ops.minmax_collect(t if t.is_contiguous() else t.contiguous(), state)


for name, part in (("first 80 MB", big[:1820]), ("last 80 MB", big[-1820:]), ("all 180 MB", big)):
    res = []
    for flush in (True, False):
        ts = []
        for _ in range(5):
            if flush:
                for o in other:
                    rd(o)            # 540 MB of other data: evicts the cache
            else:
                rd(big)              # the whole tensor streamed just before
            torch.cuda.synchronize()
            ts.append(timed(lambda: rd(part)))
        res.append(min(ts))
    nbytes = part.numel() * 4
    print(f"{name}: after other data {res[0]:.1f} us = {nbytes / res[0] / 1e6:.2f} TB/s; right after streaming the tensor {res[1]:.1f} us = {nbytes / res[1] / 1e6:.2f} TB/s", flush=True)
