"""The inverse factor U (inv(H + damp I) = U^T U) of the library against float64, with the factor's large products on the
fp16-piece kernels (default) and on the fp32 kernel (Hessian method f32), on a Hessian of the bench's activations."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from onnx_quantize_amd.hip import ops
dev = torch.device("cuda:0")
for K in (4096, 11008):
    g = torch.Generator(device=dev).manual_seed(1234 + K)
    chan = 0.1 + 3.9 * torch.rand(K, generator=g, device=dev)
    if "--outliers" in sys.argv:            # massive-activation channels: 1 % of the inputs 1000 x larger, a few tiny ones
        chan[torch.randperm(K, generator=g, device=dev)[: K // 100]] *= 1000.0
        chan[torch.randperm(K, generator=g, device=dev)[: K // 100]] *= 1e-4
    h = torch.zeros((K, K), device=dev)
    n = 0
    for _ in range(2):
        x = torch.randn((16, 2048, K), generator=g, device=dev) * chan
        n = ops.hessian_accumulate(x, h, n)
    del x
    hd = h.double()
    hd = hd + 0.01 * hd.diagonal().mean() * torch.eye(K, device=dev, dtype=torch.float64)
    ref = torch.linalg.cholesky(torch.linalg.inv(hd)).T.contiguous()
    del hd
    for method in ("auto", "f32"):
        ops.hessian_set_method(method)
        u, info = ops.gptq_factor(h, 0.01)
        d = u.double() - ref
        print(f"K={K} factor products {'fp16 pieces' if method == 'auto' else 'fp32'}: max|dU|/max|U| {float(d.abs().max() / ref.abs().max()):.3e}  "
              f"||dU||/||U|| {float(d.norm() / ref.norm()):.3e}  max_i |dU_ii|/U_ii {float((d.diagonal().abs() / ref.diagonal()).max()):.3e}  "
              f"max over rows ||dU_i||/||U_i|| {float((d.norm(dim=1) / ref.norm(dim=1)).max()):.3e}  info {int(info.item())}", flush=True)
        del u, d
    ops.hessian_set_method("auto")
    del h, ref
