"""Lab: what would 22-bit operands (two fp16 pieces, lo.lo dropped) in the trailing updates of the blocked Cholesky cost in
accuracy?  Emulated with torch on the GPU (three fp32 products of split operands), against a float64 factorisation, on a
Hessian of the bench's activation distribution.  Prints the relative error of the lower factor for both variants."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from onnx_quantize_amd.hip import ops
torch.backends.cuda.matmul.allow_tf32 = False
dev = torch.device("cuda:0")

def split22(a):
    amax = a.abs().max()
    s = torch.exp2(torch.floor(15 - torch.log2(amax)))           # power of two: max into [2^14, 2^16)
    x = a * s
    hi = x.half().float()
    lo = (x - hi).half().float()
    return hi / s, lo / s

def mm22(a, b):                                                  # a^T-free: plain a @ b with 22-bit operands, lo.lo dropped
    ah, al = split22(a)
    bh, bl = split22(b)
    return ah @ bh + (ah @ bl + al @ bh)

def blocked_chol(h, nb, mm):
    a = h.clone()
    k = a.shape[0]
    for o in range(0, k, nb):
        e = min(o + nb, k)
        l11 = torch.linalg.cholesky(a[o:e, o:e].double()).float()    # diagonal block: exact enough in either variant
        a[o:e, o:e] = l11
        if e < k:
            l21 = torch.linalg.solve_triangular(l11.double(), a[e:, o:e].double().T, upper=False).T.float()
            a[e:, o:e] = l21
            a[e:, e:] -= mm(l21, l21.T.contiguous())
    return torch.tril(a)

for K in (4096, 11008):
    g = torch.Generator(device=dev).manual_seed(1234 + K)
    chan = 0.1 + 3.9 * torch.rand(K, generator=g, device=dev)
    h = torch.zeros((K, K), device=dev)
    n = 0
    for _ in range(2):
        x = torch.randn((16, 2048, K), generator=g, device=dev) * chan
        n = ops.hessian_accumulate(x, h, n)
    del x
    hd = h.double()
    hd += 0.01 * hd.diagonal().mean() * torch.eye(K, device=dev, dtype=torch.float64)
    ref = torch.linalg.cholesky(hd)
    h32 = hd.float()
    for name, mm in (("fp32 products", lambda a, b: a @ b), ("22-bit operands", mm22)):
        l = blocked_chol(h32, 512, mm)
        err = float((l.double() - ref).abs().max() / ref.abs().max())
        rel = float(((l.double() - ref).norm() / ref.norm()))
        # what matters downstream: the inverse factor
        li = torch.linalg.solve_triangular(l.double(), torch.eye(K, device=dev, dtype=torch.float64), upper=False)
        ri = torch.linalg.solve_triangular(ref, torch.eye(K, device=dev, dtype=torch.float64), upper=False)
        ierr = float((li - ri).norm() / ri.norm())
        print(f"K={K} {name}: max|dL|/max|L| {err:.3e}  ||dL||/||L|| {rel:.3e}  ||d inv(L)||/||inv(L)|| {ierr:.3e}  cond~{float(hd.diagonal().max() / hd.diagonal().min()):.1f}", flush=True)
        del l, li, ri
    del h, hd, ref, h32
