"""First-touch cost of download destinations: plain np.empty vs mmap + MADV_HUGEPAGE vs pinned bounce + copy-out."""
import mmap
import time
import numpy as np
import torch

print("THP:", open("/sys/kernel/mm/transparent_hugepage/enabled").read().strip(), "| defrag:", open("/sys/kernel/mm/transparent_hugepage/defrag").read().strip())
torch.cuda.init()
n = 4096 * 11008
blob = torch.empty(n // 2, dtype=torch.uint8, device="cuda")
kn = torch.empty(n, dtype=torch.uint8, device="cuda")
blob.cpu(); torch.cuda.synchronize()


def huge_empty(nbytes):
    size = (nbytes + (2 << 20) - 1) // (2 << 20) * (2 << 20)
    mm = mmap.mmap(-1, size)
    mm.madvise(mmap.MADV_HUGEPAGE)
    return np.frombuffer(mm, dtype=np.uint8, count=nbytes)


def timed(fn, k=6):
    torch.cuda.synchronize(); t0 = time.perf_counter(); keep = [fn() for _ in range(k)]; torch.cuda.synchronize(); return (time.perf_counter() - t0) * 1e3 / k


def into(dst_fn, t):
    out = dst_fn(t.numel())
    torch.from_numpy(out).copy_(t)
    return out


pin = torch.empty(n, dtype=torch.uint8, pin_memory=True); pin.zero_()


def bounce(dst_fn, t):
    m = t.numel()
    pin[:m].copy_(t, non_blocking=True)
    torch.cuda.current_stream().synchronize()
    out = dst_fn(m)
    np.copyto(out, pin[:m].numpy())
    return out


for name, t in (("22.5 MB", blob), ("45 MB", kn)):
    print(name, "np.empty dst, direct D2H        ms", timed(lambda: into(lambda m: np.empty(m, np.uint8), t)))
    print(name, "hugepage dst, direct D2H        ms", timed(lambda: into(huge_empty, t)))
    print(name, "np.empty dst, pinned bounce     ms", timed(lambda: bounce(lambda m: np.empty(m, np.uint8), t)))
    print(name, "hugepage dst, pinned bounce     ms", timed(lambda: bounce(huge_empty, t)))
print("alloc only: np.empty+touch 22.5MB ms", timed(lambda: np.empty(n // 2, np.uint8).fill(0)))
print("alloc only: huge+touch 22.5MB     ms", timed(lambda: huge_empty(n // 2).fill(0)))
