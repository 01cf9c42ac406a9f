"""Hessian updates of one gemma-3-270m-shaped calibration batch (72 tapped inputs, [10, 512, 640 / 1024 / 2048]): the
grouped call against per-tensor calls on one stream and spread over four side streams.  ms per batch by wall clock over
10 batches, plus the worst deviation between the routes relative to max |H|."""
import os, sys, time, json, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from onnx_quantize_amd.hip import ops
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(2)
widths = [640, 640, 1024, 2048] * 18
xs = [torch.randn((10, 512, k), generator=g, device=dev) * (0.1 + 3.9 * torch.rand(k, generator=g, device=dev)) for k in widths]

def per_tensor(streams):
    hs = [torch.zeros((k, k), device=dev) for k in widths]
    side = [torch.cuda.Stream(device=dev) for _ in range(streams)]
    def batch(n):
        cur = torch.cuda.current_stream()
        fork = cur.record_event()
        for i, (x, h) in enumerate(zip(xs, hs)):
            s = side[i % streams] if side else cur
            s.wait_event(fork)
            with torch.cuda.stream(s):
                ops.hessian_accumulate(x, h, n)
        for s in side:
            cur.wait_stream(s)
        return n + 10
    return hs, batch

def grouped():
    hs = [torch.zeros((k, k), device=dev) for k in widths]
    def batch(n):
        return ops.hessian_accumulate_many(xs, hs, [n] * len(xs))[0]
    return hs, batch

out = {}
keep = {}
for name, make in (("per_tensor_1_stream", lambda: per_tensor(0)), ("per_tensor_4_streams", lambda: per_tensor(4)), ("grouped", grouped)):
    hs, batch = make()
    n = batch(0); n = batch(n)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        n = batch(n)
    torch.cuda.synchronize()
    out[name + "_ms_per_batch"] = round((time.perf_counter() - t0) * 100, 3)
    keep[name] = hs
dev_max = max(float((a - b).abs().max() / b.abs().max()) for a, b in zip(keep["grouped"], keep["per_tensor_1_stream"]))
out["grouped_vs_per_tensor_max_abs_over_max_h"] = dev_max
flop = sum(2.0 * 5120 * k * k for k in widths)
out["fp32_equivalent_TFLOPs_grouped"] = round(flop / out["grouped_ms_per_batch"] / 1e9, 1)
print(json.dumps(out))
