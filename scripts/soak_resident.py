#!/usr/bin/env python3
"""Soak of the ticketed RTN kernels (channel / tensor / tall groups): a few thousand back-to-back calls over mixed shapes on one
state buffer; every result must equal the first result for the same input bit for bit, and the state must come back zero.
usage: soak_resident.py [seconds]"""
import hashlib
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from onnx_quantize_amd.hip import ops  # noqa: E402

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
torch.cuda.set_device(0)
gen = torch.Generator(device="cuda").manual_seed(1)
cases = []
for (k, n) in ((4096, 11008), (11008, 4096), (4096, 4096), (512, 1028), (4100, 260), (130, 2052), (8192, 8192), (256, 512), (2048, 300)):
    w = torch.randn((k, n), generator=gen, device="cuda") * 0.05
    for qtype, strategy, g in (("int8", "channel", -1), ("int8", "tensor", -1), ("uint4", "channel", -1), ("int8", "group", 512 if k % 512 == 0 else -1)):
        if strategy == "group" and g == -1:
            continue
        cases.append((w, qtype, strategy, g))


def digest(res):
    h = hashlib.sha256()
    for t in res:
        h.update(t.cpu().numpy().tobytes())
    return h.hexdigest()[:16]


ref = [digest(ops.rtn_quantize(w, qt, st, g)) for (w, qt, st, g) in cases]
rng = torch.Generator().manual_seed(7)
t0 = time.time()
calls = checked = 0
while time.time() - t0 < budget:
    order = torch.randperm(len(cases), generator=rng).tolist()
    outs = []
    for i in order:                      # a burst of calls without any host synchronisation in between
        w, qt, st, g = cases[i]
        outs.append((i, ops.rtn_quantize(w, qt, st, g)))
        calls += 1
    for i, res in outs[::3]:
        assert digest(res) == ref[i], ("result changed", cases[i][1:], tuple(cases[i][0].shape))
        checked += 1
    state = ops._rtn_state(1, "cuda")
    torch.cuda.synchronize()
    assert int(state.count_nonzero()) == 0, "state not clean"
    print(f"{time.time() - t0:6.1f} s: {calls} calls, {checked} results compared", flush=True)
print("soak ok", calls, checked)
