"""Do the HBM-bound passes of one Hessian call hide behind another call's GEMM?  Two streams, each accumulating its own H
from its own X, against one stream doing both."""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from onnx_quantize_amd.hip import ops

dev = torch.device("cuda:0")
for K in (4096, 11008):
    T = 65536
    xs = [torch.randn(T, K, device=dev) for _ in range(2)]
    hs = [torch.zeros(K, K, device=dev) for _ in range(2)]
    st = [torch.cuda.Stream(device=dev) for _ in range(2)]
    def run(two, reps=6):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for r in range(reps):
            for i in range(2):
                with torch.cuda.stream(st[i if two else 0]):
                    ops.hessian_accumulate(xs[i], hs[i], T * (r + 1))
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / (2 * reps) * 1e3
    run(False, 1); run(True, 1)
    a = run(False); b = run(True); a2 = run(False); b2 = run(True)
    print(json.dumps({"K": K, "ms_per_call_one_stream": [round(a, 3), round(a2, 3)], "ms_per_call_two_streams": [round(b, 3), round(b2, 3)]}), flush=True)
