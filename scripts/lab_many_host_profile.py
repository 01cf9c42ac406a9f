import cProfile, pstats, sys, os, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from onnx_quantize_amd.hip import ops
dev = torch.device("cuda", 0)
shapes = ([(640, 1024)] + [(640, 256)] * 2 + [(1024, 640)] + [(640, 2048)] * 2 + [(2048, 640)]) * 18
ws = [torch.randn(s, device=dev) * 0.02 for s in shapes]
for _ in range(3):
    r = ops.rtn_quantize_many(ws, "int8", 128, layout="kn")
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20):
    r = ops.rtn_quantize_many(ws, "int8", 128, layout="kn")
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"host issue {1e3 * (t1 - t0) / 20:.3f} ms per call, total incl sync {1e3 * (t2 - t0) / 20:.3f}")
pr = cProfile.Profile(); pr.enable()
for _ in range(20):
    r = ops.rtn_quantize_many(ws, "int8", 128, layout="kn")
pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
