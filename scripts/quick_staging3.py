"""Cold vs warm pageable transfers, and overlap of a worker thread's uploads with the main thread's downloads."""
import threading
import time
import numpy as np
import torch

torch.cuda.init()
base = np.random.default_rng(0).standard_normal((4096, 11008), dtype=np.float32)
n = base.size
blob = torch.empty(n // 2, dtype=torch.uint8, device="cuda")
kn = torch.empty(n, dtype=torch.uint8, device="cuda")
torch.from_numpy(base).cuda(); torch.cuda.synchronize()


def fresh(k=6):
    return [base.copy() for _ in range(k)]


def timed(fn):
    torch.cuda.synchronize(); t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); return (time.perf_counter() - t0) * 1e3 / 6


ws = fresh()
print("cold pageable .cuda()           ms/weight", timed(lambda: [torch.from_numpy(w).cuda() for w in ws]))
print("warm pageable .cuda()           ms/weight", timed(lambda: [torch.from_numpy(w).cuda() for w in ws]))
print("D2H .cpu() 22.5 MB (fresh dst)  ms/each  ", timed(lambda: [blob.cpu() for _ in range(6)]))
print("D2H .cpu() 45 MB (fresh dst)    ms/each  ", timed(lambda: [kn.cpu() for _ in range(6)]))


def chunked(t, chunk=16 << 20):
    out = np.empty(t.numel(), np.uint8)
    d = torch.from_numpy(out)
    for o in range(0, t.numel(), chunk):
        d[o:o + chunk].copy_(t[o:o + chunk])
    return out


print("D2H chunked direct 45 MB        ms/each  ", timed(lambda: [chunked(kn) for _ in range(6)]))
print("D2H chunked direct 22.5 MB      ms/each  ", timed(lambda: [chunked(blob) for _ in range(6)]))
for ch in (4, 8, 32):
    print(f"D2H chunked {ch} MB pieces, 45 MB   ms/each  ", timed(lambda: [chunked(kn, ch << 20) for _ in range(6)]))
side = torch.cuda.Stream()


def overlap(ws, register):
    rt = torch.cuda.cudart()
    evs = [None] * 6
    devs = [None] * 6
    cv = threading.Condition()

    def work():
        torch.cuda.set_device(0)
        for i, w in enumerate(ws):
            if register:
                rt.cudaHostRegister(w.ctypes.data, w.nbytes, 0)
            with torch.cuda.stream(side):
                d = torch.from_numpy(w).to("cuda", non_blocking=True)
                e = torch.cuda.Event(); e.record(side)
            with cv:
                evs[i], devs[i] = e, d
                cv.notify_all()
    th = threading.Thread(target=work); th.start()
    for i in range(6):
        with cv:
            while evs[i] is None:
                cv.wait()
        torch.cuda.current_stream().wait_event(evs[i])
        chunked(blob)
    th.join()
    torch.cuda.synchronize()
    if register:
        for w in ws:
            rt.cudaHostUnregister(w.ctypes.data)


ws = fresh(); print("worker uploads (cold, pageable) + main D2H 22.5 MB   ms/weight", timed(lambda: overlap(ws, False)))
ws = fresh(); print("worker uploads (cold, registered) + main D2H 22.5 MB ms/weight", timed(lambda: overlap(ws, True)))
ws = fresh(); print("serial: cold upload then D2H 22.5 MB                 ms/weight", timed(lambda: [(torch.from_numpy(w).cuda(), chunked(blob)) for w in ws]))
