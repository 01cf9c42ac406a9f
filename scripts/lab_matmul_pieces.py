#!/usr/bin/env python3
"""`ops.matmul_pieces` (fp16 two-piece operands on the matrix cores) against torch's fp32 matmul: time and error vs float64.

    python scripts/lab_matmul_pieces.py [--shapes 16384x4096x11008,16384x11008x4096,16384x4096x4096,2048x640x2048]
"""
import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from onnx_quantize_amd.hip import ops  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--shapes", default="16384x4096x11008,16384x11008x4096,16384x4096x4096,2048x640x2048,256x1024x1024")
    a = ap.parse_args()
    for shp in a.shapes.split(","):
        t, k, n = (int(v) for v in shp.split("x"))
        gen = torch.Generator(device="cuda").manual_seed(t + k + n)
        x = torch.randn(t, k, generator=gen, device="cuda") * torch.rand(k, generator=gen, device="cuda") * 3
        w = torch.randn(k, n, generator=gen, device="cuda") / k ** 0.5
        wp = ops.matmul_prepare(w, False)

        def timed(fn, reps=5):
            fn()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(reps):
                out = fn()
            torch.cuda.synchronize()
            return (time.perf_counter() - t0) / reps, out
        t_torch, y_torch = timed(lambda: x @ w)
        t_pieces, y_pieces = timed(lambda: ops.matmul_pieces(x, wp))
        rows = slice(0, min(t, 512))
        ref = (x[rows].double() @ w.double())
        err = lambda y: ((y[rows].double() - ref).norm() / ref.norm()).item()      # noqa: E731
        flops = 2.0 * t * k * n
        print(json.dumps({"shape": shp, "torch_ms": round(t_torch * 1e3, 3), "pieces_ms": round(t_pieces * 1e3, 3),
                          "torch_tflops": round(flops / t_torch / 1e12, 1), "pieces_fp32_equiv_tflops": round(flops / t_pieces / 1e12, 1),
                          "err_torch_vs_f64": err(y_torch), "err_pieces_vs_f64": err(y_pieces)}), flush=True)


if __name__ == "__main__":
    main()
