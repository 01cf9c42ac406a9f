"""Cost of page-locking a fresh 180 MB NumPy array in place (hipHostRegister / Unregister) against its upload."""
import time, json, numpy as np, torch
rt = torch.cuda.cudart()
w = np.random.default_rng(0).standard_normal((4096, 11008), dtype=np.float32)
torch.cuda.init(); torch.zeros(1, device="cuda")
out = []
for rep in range(5):
    a = w.copy()
    t0 = time.perf_counter(); rc = int(rt.cudaHostRegister(a.ctypes.data, a.nbytes, 0)); t1 = time.perf_counter()
    d = torch.from_numpy(a).to("cuda", non_blocking=True); torch.cuda.synchronize(); t2 = time.perf_counter()
    rt.cudaHostUnregister(a.ctypes.data); t3 = time.perf_counter()
    b = w.copy()
    t4 = time.perf_counter(); d2 = torch.from_numpy(b).to("cuda"); torch.cuda.synchronize(); t5 = time.perf_counter()
    out.append({"register_ms": round((t1 - t0) * 1e3, 2), "async_copy_ms": round((t2 - t1) * 1e3, 2), "unregister_ms": round((t3 - t2) * 1e3, 2),
                "pageable_copy_ms": round((t5 - t4) * 1e3, 2), "rc": rc})
print(json.dumps(out))
