import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from onnx_quantize_amd.hip import ops
K = int(sys.argv[1]) if len(sys.argv) > 1 else 11008
T = int(sys.argv[2]) if len(sys.argv) > 2 else 8192
x = torch.randn((8, T // 8, K), device="cuda")
h = torch.zeros((K, K), device="cuda")
for _ in range(4):
    ops.hessian_accumulate(x, h, 0)
torch.cuda.synchronize()
