"""Error of the X^T X kernels against float64 at production length (T = 4 x 65536 rows, K = 4096, bench-like data):
a 256-column strip of H is recomputed in float64.  python scripts/quick_hess_accuracy.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from onnx_quantize_amd.hip import ops

dev = torch.device("cuda", 0)
k, seqs, seq = 4096, 32, 2048
for offset in (0.0, 0.5):
    g = torch.Generator(device=dev).manual_seed(7)
    chan = 0.1 + 3.9 * torch.rand(k, generator=g, device=dev)
    batches = [torch.randn((seqs, seq, k), generator=g, device=dev) * chan + offset for _ in range(4)]
    ref = torch.zeros((256, k), dtype=torch.float64, device=dev)
    for x in batches:
        x2 = x.reshape(-1, k)
        for i in range(0, x2.shape[0], 16384):
            blk = x2[i:i + 16384].double()
            ref += blk[:, :256].t() @ blk
    ref *= 2.0 / (4 * seqs)
    for m in ("f32", "bf16x6", "bf16x9"):
        ops.hessian_set_method(m)
        h = torch.zeros((k, k), device=dev)
        n = 0
        for x in batches:
            n = ops.hessian_accumulate(x, h, n)
        err = (h[:256].double() - ref).abs()
        print(f"offset={offset} {m:7s} n={n} max|err|/max|H| = {float(err.max() / ref.abs().max()):.3e}   "
              f"median rel = {float((err / ref.abs().clamp_min(1e-30)).median()):.3e}   sym = {float((h - h.t()).abs().max()):.1e}", flush=True)
