"""One batched factor chain (the bench's shape: 8 Hessians of 11008 or 24 of 4096) for a rocprofv3 kernel trace."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from onnx_quantize_amd.hip import ops
K, B = int(sys.argv[1]), int(sys.argv[2])
x = torch.randn((16, 1024, K), device="cuda") * (0.1 + 3.9 * torch.rand(K, device="cuda"))
h = torch.zeros((K, K), device="cuda")
ops.hessian_accumulate(x, h, 0)
hs = h.unsqueeze(0).repeat(B, 1, 1).contiguous()
del x
ops.gptq_factor_batched(hs, 0.01, True)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
u, info = ops.gptq_factor_batched(hs, 0.01, True)
e1.record(); torch.cuda.synchronize()
print("K", K, "B", B, "ms", e0.elapsed_time(e1), "info", info.tolist()[:2])
