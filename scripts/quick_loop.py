import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from onnx_quantize_amd.hip import ops
K, N = 4096, 4096
x = torch.randn((8, 1024, K), device="cuda")
h = torch.zeros((K, K), device="cuda")
ops.hessian_accumulate(x, h, 0)
u, info = ops.gptq_factor(h, 0.01)
w = torch.randn((K, N), device="cuda") * 0.02
_, s0, z0 = ops.rtn_quantize(w, "int4", "channel", -1, emit_q=False)
for _ in range(3):
    ops.gptq_loop(w.clone(), u, "int4", 128, False, False, 1.0, False, 128, "corrected", s0, z0)
torch.cuda.synchronize()
