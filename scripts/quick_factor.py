import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from onnx_quantize_amd.hip import ops
K = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
x = torch.randn((8, 1024, K), device="cuda")
h = torch.zeros((K, K), device="cuda")
ops.hessian_accumulate(x, h, 0)
for _ in range(3):
    u, info = ops.gptq_factor(h, 0.01)
torch.cuda.synchronize()
print("info", int(info.item()))
