"""Which host calls of the corrected whole-model pass take long (allocator / driver stalls)?  Wraps torch.empty / zeros."""
import os, sys, time, json, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench_gptq
slow = []
def wrap(name):
    orig = getattr(torch, name)
    def f(*a, **k):
        t0 = time.perf_counter()
        r = orig(*a, **k)
        dt = time.perf_counter() - t0
        if dt > 5e-3:
            slow.append((round(dt * 1e3, 1), name, tuple(r.shape), str(r.dtype)))
        return r
    setattr(torch, name, f)
for n in ("empty", "zeros", "empty_like", "cat"):
    wrap(n)
args = bench_gptq.build_parser().parse_args(sys.argv[1:] or ["--no-cpu-baseline", "--extra-passes", "corrected", "--hessian-methods", "auto"])
# the ops the pass calls, timed on the host: a call that blocks (driver allocation, implicit synchronisation) shows here
from onnx_quantize_amd.hip import ops
def wrap_op(obj, name):
    orig = getattr(obj, name)
    def f(*a, **k):
        t0 = time.perf_counter()
        r = orig(*a, **k)
        dt = time.perf_counter() - t0
        if dt > 5e-3:
            slow.append((round(dt * 1e3, 1), "ops." + name, round(t0 % 1000, 3)))
        return r
    setattr(obj, name, f)
for n in ("gptq_shared_factors", "gptq_quantize", "pack_nibbles", "hessian_accumulate"):
    wrap_op(ops, n)
wrap_op(ops.HessianPipeline, "accumulate")
from bench import init_ranks
dev, rank, world = init_ranks(1)
res = bench_gptq.run(args, dev, rank, world)
print(json.dumps({"parity": res["seconds"], "corrected": (res.get("corrected") or {}).get("seconds")}))
print("slow allocations (ms, fn, shape):")
for s in slow: print("  ", s)
stats = torch.cuda.memory_stats()
print({k: stats[k] for k in ("num_alloc_retries", "num_ooms", "reserved_bytes.all.peak", "allocated_bytes.all.peak", "num_device_alloc", "num_device_free") if k in stats})
