import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from onnx_quantize_amd.hip import ops
def ev(fn, iters=3, warm=1):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters
for K, T in ((4096, 16384), (11008, 8192)):
    x = torch.randn((8, T // 8, K), device="cuda")
    h = torch.zeros((K, K), device="cuda")
    ms = ev(lambda: ops.hessian_accumulate(x, h, 0))
    print(f"hessian K={K} T={T}: {ms:.2f} ms  {2*T*K*K/ms/1e9:.1f} TFLOP/s (2TK^2 convention), {T*K*K/ms/1e9:.1f} TFLOP/s executed", flush=True)
    ops.hessian_accumulate(x, h, 0)
    ms = ev(lambda: ops.gptq_factor(h, 0.01), iters=2)
    print(f"factor  K={K}: {ms:.2f} ms", flush=True)
    u, info = ops.gptq_factor(h, 0.01)
    print("info", int(info.item()))
    for N in ((4096, 11008) if K == 4096 else (4096,)):
        w = torch.randn((K, N), device="cuda") * 0.02
        for mode in ("parity", "corrected"):
            _, s0, z0 = ops.rtn_quantize(w, "int4", "channel", -1, emit_q=False)
            ms = ev(lambda: ops.gptq_loop(w.clone(), u, "int4", 128, False, False, 1.0, False, 128, mode, s0, z0), iters=2)
            print(f"loop    K={K} N={N} {mode}: {ms:.2f} ms", flush=True)
        ms = ev(lambda: ops.gptq_quantize(w, h, "int4", "group", 128), iters=2)
        print(f"gptq_quantize total (parity) K={K} N={N}: {ms:.2f} ms -> {K*N/ms/1e3:.1f} M-param/s", flush=True)
