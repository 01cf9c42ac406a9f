#!/usr/bin/env python3
"""Lab: does faulting in a download's destination from four threads (staging._prefault) help or hurt the SINGLE-weight seam
(bench.py's `seam` object: 12 weights of 4096 x 11008, uint4 g128 -> MatMulNBits arrays)?  Alternates the two settings in one
process, five passes each, and splits a pass into upload / kernel / download.
usage: lab_seam_prefault.py"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import _Value  # noqa: E402
from onnx_quantize_amd import QConfig, QuantType, QWeightArgs, seam, staging  # noqa: E402

torch.cuda.set_device(0)
w_host = np.random.default_rng(0).standard_normal((4096, 11008), dtype=np.float32)
qc = QConfig(weights=QWeightArgs(dtype=QuantType.QUInt4, group_size=128))
seam.weight_arrays(_Value("warm", w_host.copy()), qc, None, True)
timers = {"upload": 0.0, "download": 0.0}
up0, down0 = staging.upload, staging.download


def timed(name, fn):
    def run(*a, **k):
        t0 = time.perf_counter()
        r = fn(*a, **k)
        timers[name] += time.perf_counter() - t0
        return r
    return run


staging.upload, staging.download = timed("upload", up0), timed("download", down0)
for rnd in range(5):
    for label, thr in (("prefault>=4MiB", 4 << 20), ("prefault off", 1 << 40)):
        staging._PREFAULT_MIN_BYTES = thr
        mats = [w_host.copy() for _ in range(12)]
        timers["upload"] = timers["download"] = 0.0
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i, w in enumerate(mats):
            last = seam.weight_arrays(_Value(f"w{i}", w), qc, None, True)
        t = time.perf_counter() - t0
        print(f"round {rnd} {label:15s} {t * 1e3 / 12:6.2f} ms per weight   upload {timers['upload'] * 1e3 / 12:5.2f}  download {timers['download'] * 1e3 / 12:5.2f}", flush=True)
