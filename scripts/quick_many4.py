import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from onnx_quantize_amd.hip import ops
gen = torch.Generator(device="cuda").manual_seed(5)
shapes = [(4096, 4096)] * 4 + [(4096, 11008)] * 2 + [(11008, 4096)]
base = {sh: torch.randn(sh, generator=gen, device="cuda") * 0.02 for sh in set(shapes)}
ws = [base[sh].clone() for _ in range(32) for sh in shapes]
for rep in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    res = ops.rtn_quantize_many(ws, "uint4", 128, layout="nbits")
    e1.record()
    t1 = time.perf_counter()
    torch.cuda.synchronize(); t2 = time.perf_counter()
    print(f"rep {rep}: host enqueue {1e3 * (t1 - t0):.2f} ms, total wall {1e3 * (t2 - t0):.2f} ms, device {e0.elapsed_time(e1):.2f} ms", flush=True)
    del res
# same-shape groups in model order vs grouped: what the kernels alone take
for name, lst in (("4096x4096 only", [w for w in ws if w.shape == (4096, 4096)]), ("4096x11008 only", [w for w in ws if w.shape == (4096, 11008)]),
                  ("11008x4096 only", [w for w in ws if w.shape == (11008, 4096)])):
    ops.rtn_quantize_many(lst, "uint4", 128, layout="nbits"); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); r = ops.rtn_quantize_many(lst, "uint4", 128, layout="nbits"); e1.record(); torch.cuda.synchronize()
    print(name, len(lst), f"{e0.elapsed_time(e1):.3f} ms, {e0.elapsed_time(e1) / len(lst) * 1e3:.2f} us per matrix")
    del r
