"""Corrected-mode GPTQ loop alone on the three Llama-2-7B shapes: ms per call (HIP events), old and new kernel by env."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from onnx_quantize_amd.hip import ops
for K, N in ((4096, 4096), (4096, 11008), (11008, 4096)):
    x = torch.randn((8, 1024, K), device="cuda")
    h = torch.zeros((K, K), device="cuda")
    ops.hessian_accumulate(x, h, 0)
    u, info = ops.gptq_factor(h, 0.01)
    w = torch.randn((K, N), device="cuda") * 0.02
    _, s0, z0 = ops.rtn_quantize(w, "int4", "channel", -1, emit_q=False)
    ws = [w.clone() for _ in range(4)]
    ops.gptq_loop(ws[0], u, "int4", 128, False, False, 1.0, False, 128, "corrected", s0, z0)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(3):
        ops.gptq_loop(ws[1 + i], u, "int4", 128, False, False, 1.0, False, 128, "corrected", s0, z0)
    e1.record()
    torch.cuda.synchronize()
    print(f"K={K} N={N} rows16={os.environ.get('OQ_GPTQ_ROWS16', '1')}: {e0.elapsed_time(e1) / 3:.3f} ms per loop", flush=True)
    del x, h, u, w, ws
