#!/usr/bin/env python3
"""`ops.matmul_pieces` at the edges: all-zero operands, one huge element next to a row 10^10 times smaller (what a per-operand scale
costs), operands at 1e-30 / 1e30 (the whole fp32 exponent range must work)."""
import sys; sys.path.insert(0, "/root/repo")
import torch
from onnx_quantize_amd.hip import ops
w = torch.randn(1024, 768, device="cuda") / 32
z = torch.zeros(300, 1024, device="cuda")
print("zeros:", ops.matmul_pieces(z, w).abs().max().item())
x = torch.randn(300, 1024, device="cuda")
print("zero weight:", ops.matmul_pieces(x, torch.zeros_like(w)).abs().max().item())
x[5, 7] = 3e4; x[9] *= 1e-6
ref = x.double() @ w.double()
for name, y in (("pieces", ops.matmul_pieces(x, w)), ("torch", x @ w)):
    err_rows = ((y.double() - ref).norm(dim=1) / ref.norm(dim=1))
    print(name, "worst row rel err", err_rows.max().item(), "tiny row", err_rows[9].item())
tiny = torch.randn(300, 1024, device="cuda") * 1e-30
y = ops.matmul_pieces(tiny, w); r = tiny.double() @ w.double()
print("1e-30 scale rel err", ((y.double() - r).norm() / r.norm()).item())
big = torch.randn(300, 1024, device="cuda") * 1e30
y = ops.matmul_pieces(big, w); r = big.double() @ w.double()
print("1e30 scale rel err", ((y.double() - r).norm() / r.norm()).item())
