#!/usr/bin/env python3
"""Lab: `staging.download` into FRESH destinations that are all held (a model run keeps every result until the file is written),
with and without the four-thread prefault: 120 results of 22.5 MB (a 4096 x 11008 uint4 blob), alternating settings."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from onnx_quantize_amd import staging  # noqa: E402

torch.cuda.set_device(0)
t = torch.randint(0, 255, (11008, 32, 64), dtype=torch.uint8, device="cuda")
staging.download(t)
for rnd in range(3):
    for label, thr in (("prefault >= 4 MiB", 4 << 20), ("prefault off", 1 << 40)):
        staging._PREFAULT_MIN_BYTES = thr
        held = []
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(120):
            held.append(staging.download(t))
        dt = time.perf_counter() - t0
        print(f"round {rnd} {label:18s} {dt * 1e3 / 120:6.2f} ms per result = {t.numel() * 120 / dt / 1e9:5.1f} GB/s", flush=True)
        del held
