"""Accuracy (against float64) and speed of the three X^T X kernels.  python scripts/quick_hess_methods.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from onnx_quantize_amd.hip import ops

dev = torch.device("cuda", 0)


def accuracy(t, k, seed, heavy=False):
    g = torch.Generator(device=dev).manual_seed(seed)
    x = torch.randn((t, k), generator=g, device=dev)
    if heavy:
        x = x * (0.1 + 3.9 * torch.rand(k, generator=g, device=dev)) + 0.5
    ref = (x.double().t() @ x.double()) * (2.0 / 8)
    out = {}
    for m in ("f32", "bf16x6", "bf16x9"):
        ops.hessian_set_method(m)
        h = torch.zeros((k, k), device=dev)
        ops.hessian_accumulate(x.reshape(8, t // 8, k), h, 0)
        err = (h.double() - ref).abs()
        out[m] = (float(err.max() / ref.abs().max()), float((err / ref.abs().clamp_min(1e-30)).median()),
                  float((h - h.t()).abs().max()))
    return out


for (t, k, heavy) in ((4096, 1024, False), (16384, 2048, True), (8192, 1280, True), (1000 * 8, 1026, False)):
    print(f"T={t} K={k} heavy={heavy}")
    for m, (emax, emed, asym) in accuracy(t, k, 1, heavy).items():
        print(f"   {m:7s} max|err|/max|H| = {emax:.3e}   median rel err = {emed:.3e}   asymmetry = {asym:.1e}")

for k in (4096, 11008):
    t = 65536
    x = torch.randn((32, 2048, k), device=dev) * (0.1 + 3.9 * torch.rand(k, device=dev))
    for m in ("f32", "bf16x6", "bf16x9"):
        ops.hessian_set_method(m)
        h = torch.zeros((k, k), device=dev)
        n = ops.hessian_accumulate(x, h, 0)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 3
        e0.record()
        for _ in range(reps):
            n = ops.hessian_accumulate(x, h, n)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / reps
        print(f"K={k} T={t} {m:7s} {ms:8.2f} ms   {2.0 * t * k * k / ms / 1e9:8.1f} TFLOP/s full-matrix equivalent,"
              f" {t * k * (k + 256) / ms / 1e9:8.1f} executed (fp32-equivalent)")
