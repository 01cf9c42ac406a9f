"""Is an H2D copy from hipHostRegister'ed NumPy memory asynchronous, and does it overlap with D2H + kernels?"""
import threading
import time
import numpy as np
import torch

torch.cuda.init()
rt = torch.cuda.cudart()
ws = [np.random.default_rng(i).standard_normal((4096, 11008), dtype=np.float32) for i in range(6)]
n = ws[0].size
side = torch.cuda.Stream()
blob = torch.empty(n // 2, dtype=torch.uint8, device="cuda")


def sync_time(fn):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    fn()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    return (t1 - t0) * 1e3, (time.perf_counter() - t0) * 1e3


for w in ws:
    assert int(rt.cudaHostRegister(w.ctypes.data, n * 4, 0)) == 0
src = [torch.from_numpy(w) for w in ws]
print("is_pinned after register:", src[0].is_pinned())
devs = [torch.empty((4096, 11008), device="cuda") for _ in ws]
print("registered non_blocking copy: call / total ms", sync_time(lambda: devs[0].copy_(src[0], non_blocking=True)))
print("unregistered (fresh array) non_blocking:      ", sync_time(lambda: devs[0].copy_(torch.from_numpy(ws[0].copy()), non_blocking=True)))


def consume(i):   # stand-in for kernel + download of weight i
    blob.cpu()


def run_async():
    evs = []
    with torch.cuda.stream(side):
        for i in range(6):
            devs[i].copy_(src[i], non_blocking=True)
            e = torch.cuda.Event(); e.record(side); evs.append(e)
    for i in range(6):
        torch.cuda.current_stream().wait_event(evs[i])
        consume(i)


print("6 x (async H2D on side stream, D2H 22.5 MB on main): issue / total ms", sync_time(run_async))


def run_serial():
    for i in range(6):
        devs[i].copy_(src[i])
        consume(i)


print("6 x serial (H2D then D2H):                          ", sync_time(run_serial))


def run_thread():
    evs = [None] * 6
    cv = threading.Condition()

    def work():
        torch.cuda.set_device(0)
        for i in range(6):
            with torch.cuda.stream(side):
                devs[i].copy_(torch.from_numpy(ws[i]), non_blocking=True)
                e = torch.cuda.Event(); e.record(side)
            with cv:
                evs[i] = e
                cv.notify_all()
    th = threading.Thread(target=work); th.start()
    for i in range(6):
        with cv:
            while evs[i] is None:
                cv.wait()
        torch.cuda.current_stream().wait_event(evs[i])
        consume(i)
    th.join()


print("6 x worker thread uploads + main D2H:               ", sync_time(run_thread))
for w in ws:
    rt.cudaHostUnregister(w.ctypes.data)
print("after unregister, worker thread variant again:      ", sync_time(run_thread))
print("D2H 45 MB chunked 16 MB:", sync_time(lambda: [torch.empty(16 << 20, dtype=torch.uint8).copy_(torch.empty(16 << 20, dtype=torch.uint8, device='cuda')) for _ in range(3)]))
