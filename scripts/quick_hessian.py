"""Hessian methods side by side on one MI355X: time per call and error against float64 (K = 4096 / 11008, 65 536 rows)."""
import sys, time, json
import numpy as np, torch
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from onnx_quantize_amd.hip import ops

dev = torch.device("cuda:0")
torch.manual_seed(0)
out = []
METHODS = sys.argv[1].split(",") if len(sys.argv) > 1 else ["f32", "bf16x6", "bf16x9", "f16x3"]
SHAPES = [(int(a), 65536) for a in sys.argv[2].split(",")] if len(sys.argv) > 2 else [(4096, 65536), (11008, 65536)]
for K, T in SHAPES:
    X = torch.randn(T, K, device=dev, dtype=torch.float32)
    X[:, : K // 8] *= 30.0          # outlier channels, like LLM activations
    Xs = X[:8192].double()
    ref = (2.0 / 8192) * (Xs.T @ Xs)
    for m in METHODS:
        ops.hessian_set_method(m)
        H = torch.zeros(K, K, device=dev, dtype=torch.float32)
        n = ops.hessian_accumulate(X[:8192], H, 0)
        err = float((H.double() - ref).abs().max() / ref.abs().max())
        H.zero_()
        for _ in range(2): ops.hessian_accumulate(X, H, T)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        reps = 5
        for _ in range(reps): ops.hessian_accumulate(X, H, T)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / reps * 1e3
        rec = {"K": K, "T": T, "method": m, "ms": round(ms, 3), "tflops_fp32_equiv": round(2.0 * T * K * K / ms / 1e9, 1), "rel_err_vs_f64": err}
        print(json.dumps(rec), flush=True)
        out.append(rec)
    del X, Xs, ref
ops.hessian_set_method("auto")
