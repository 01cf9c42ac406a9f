"""Host <-> HBM transfer costs on the GPU box for one 4096x11008 fp32 weight (180 MB) and its 22.5 MB blob."""
import time
import numpy as np
import torch

torch.cuda.init()
w = np.random.default_rng(0).standard_normal((4096, 11008), dtype=np.float32)
n = w.size


def t(fn, reps=5, warm=1):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) * 1e3)
    return min(ts), sorted(ts)[len(ts) // 2]


print("pageable from_numpy().cuda()            ms min/med", t(lambda: torch.from_numpy(w).cuda()))
t0 = time.perf_counter(); pin = torch.empty(n, dtype=torch.float32, pin_memory=True); print("pinned alloc 180 MB ms", (time.perf_counter() - t0) * 1e3)
t0 = time.perf_counter(); pin2 = torch.empty(n, dtype=torch.float32, pin_memory=True); print("pinned alloc again ms", (time.perf_counter() - t0) * 1e3)
src = torch.from_numpy(w).reshape(-1)
print("memcpy into pinned (torch copy_)         ms", t(lambda: pin.copy_(src)))
for th in (1, 4, 16):
    torch.set_num_threads(th)
    print(f"  ... with torch threads={th}             ms", t(lambda: pin.copy_(src)))
torch.set_num_threads(16)
pn = pin.numpy()
print("memcpy into pinned (np.copyto)           ms", t(lambda: np.copyto(pn, w.reshape(-1))))
dev = torch.empty(n, dtype=torch.float32, device="cuda")
print("pinned -> HBM (non_blocking)             ms", t(lambda: dev.copy_(pin, non_blocking=True)))
rt = torch.cuda.cudart()
ptr = w.ctypes.data


def reg_copy():
    r = rt.cudaHostRegister(ptr, n * 4, 0)
    assert int(r) == 0, r
    v = torch.from_numpy(w).reshape(-1)
    dev.copy_(v, non_blocking=True)
    torch.cuda.synchronize()
    rt.cudaHostUnregister(ptr)


try:
    print("hostRegister + H2D + unregister          ms", t(reg_copy))
    t0 = time.perf_counter(); rt.cudaHostRegister(ptr, n * 4, 0); print("  register alone ms", (time.perf_counter() - t0) * 1e3)
    v = torch.from_numpy(w).reshape(-1)
    print("  registered -> HBM                      ms", t(lambda: dev.copy_(v, non_blocking=True)))
    t0 = time.perf_counter(); rt.cudaHostUnregister(ptr); print("  unregister alone ms", (time.perf_counter() - t0) * 1e3)
except Exception as e:
    print("hostRegister failed:", e)
blob = torch.empty(n // 2, dtype=torch.uint8, device="cuda")
kn = torch.empty(n, dtype=torch.uint8, device="cuda")
print("D2H pageable 22.5 MB (.cpu())            ms", t(lambda: blob.cpu()))
print("D2H pageable 45 MB (.cpu())              ms", t(lambda: kn.cpu()))
pb = torch.empty(n // 2, dtype=torch.uint8, pin_memory=True)
print("D2H into pinned 22.5 MB                  ms", t(lambda: pb.copy_(blob, non_blocking=True)))
out = np.empty(n // 2, np.uint8)
print("pinned -> numpy memcpy 22.5 MB           ms", t(lambda: np.copyto(out, pb.numpy())))
print("np.empty + D2H direct into numpy 22.5 MB ms", t(lambda: torch.from_numpy(np.empty(n // 2, np.uint8)).copy_(blob)))
