#!/usr/bin/env python3
"""Threaded soak of the RTN entry points: N threads, each on its own stream, issue bursts of ticketed calls (per channel / per
tensor / tall groups: chained through the library's event, rtn_resident.hip::TicketChain) mixed with fused group calls (blob,
[K,N]; free to overlap); every result must equal the single-thread result bit for bit.   usage: soak_threads.py [seconds] [threads]"""
import hashlib
import os
import sys
import threading
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from onnx_quantize_amd.hip import ops  # noqa: E402

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 30.0
nthreads = int(sys.argv[2]) if len(sys.argv) > 2 else 3
torch.cuda.set_device(0)
gen = torch.Generator(device="cuda").manual_seed(3)
cases = []
for (k, n) in ((4096, 4096), (4096, 1024), (8192, 512), (2048, 2048), (512, 1028), (12288, 256)):
    w = torch.randn((k, n), generator=gen, device="cuda") * 0.05
    cases += [(w, "int8", "channel", -1, "kn"), (w, "uint8", "tensor", -1, "kn"), (w, "uint4", "group", 128, "nbits"), (w, "int4", "group", 128, "kn")]
    if k % 1024 == 0:
        cases.append((w, "int8", "group", 1024, "kn"))


def digest(res):
    h = hashlib.sha256()
    for t in res:
        h.update(t.cpu().numpy().tobytes())
    return h.hexdigest()[:16]


ref = [digest(ops.rtn_quantize(w, qt, st, g, layout=lay)) for (w, qt, st, g, lay) in cases]
torch.cuda.synchronize()
errors, counts = [], [0] * nthreads
t0 = time.time()


def worker(tid):
    try:
        torch.cuda.set_device(0)
        stream = torch.cuda.Stream()
        rng = torch.Generator().manual_seed(100 + tid)
        with torch.cuda.stream(stream):
            while time.time() - t0 < budget and not errors:
                order = torch.randperm(len(cases), generator=rng).tolist()
                outs = []
                for i in order:
                    w, qt, st, g, lay = cases[i]
                    outs.append((i, ops.rtn_quantize(w, qt, st, g, layout=lay)))
                    counts[tid] += 1
                stream.synchronize()
                for i, res in outs[::4]:
                    if digest(res) != ref[i]:
                        errors.append((tid, cases[i][1:], tuple(cases[i][0].shape)))
    except Exception as e:  # noqa: BLE001
        errors.append((tid, repr(e)))


threads = [threading.Thread(target=worker, args=(t,)) for t in range(nthreads)]
for t in threads:
    t.start()
while any(t.is_alive() for t in threads):
    time.sleep(5)
    print(f"{time.time() - t0:6.1f} s: calls per thread {counts}", flush=True)
    if time.time() - t0 > budget + 60:
        print("threads still alive long after the budget: giving up", flush=True)
        os._exit(3)
print("errors:", errors)
print("soak ok" if not errors else "SOAK FAILED", sum(counts))
sys.exit(1 if errors else 0)
