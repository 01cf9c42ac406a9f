import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from onnx_quantize_amd.hip import ops
dev = torch.device("cuda:0")
def t(label):
    big = torch.randn((64, 2048, 2560), device=dev)
    st = ops.minmax_state(dev)
    for _ in range(3): ops.minmax_collect(big, st)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): ops.minmax_collect(big, st)
    e1.record(); torch.cuda.synchronize()
    print(label, e0.elapsed_time(e1) * 1e3 / 20, "us", flush=True)
    del big
t("fresh")
xs = [torch.randn((10, 512, k), device=dev) for k in (640, 1024, 2048)]
hs = [torch.zeros((x.shape[-1],) * 2, device=dev) for x in xs]
for x, h in zip(xs, hs): ops.hessian_accumulate(x, h, 0)
torch.cuda.synchronize(); t("after per-tensor hessians")
ops.hessian_accumulate_many(xs, hs, [10, 10, 10])
torch.cuda.synchronize(); t("after grouped hessians")
torch.cuda.empty_cache(); t("after empty_cache")
