#!/usr/bin/env python3
"""Lab: the [K,N] layouts through `ops.rtn_quantize` (stateful entry point: parameters transposed inside the launch) against the
plain entry point (`oq_rtn_quantize_f32`: staged + a transpose launch): bytes equal, time per call.
usage: lab_kn_inlaunch.py [lib.so]"""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from onnx_quantize_amd.hip import _lib as L  # noqa: E402

if len(sys.argv) > 1:
    L.LIB_PATH = os.path.abspath(sys.argv[1])
from onnx_quantize_amd.hip import ops  # noqa: E402

torch.cuda.set_device(0)
lib = L.load()
CASES = [(4096, 11008, "uint4", 128, "kn"), (4096, 11008, "int8", 128, "kn"), (4096, 27648, "int8", 128, "kn"), (4096, 32000, "uint4", 128, "kn"),
         (8192, 11008, "int8", 128, "kn"), (4096, 5376, "int8", 128, "kn"), (4096, 11008, "int4", 128, "kn_packed4"), (8192, 11008, "int4", 128, "kn_packed4"),
         (4096, 27648, "int4", 128, "kn_packed4"), (4096, 32000, "int4", 128, "kn_packed4"), (4096, 5376, "int4", 128, "kn_packed4"), (4096, 11000, "int4", 128, "kn_packed4")]
for (k, n, qtype, g, layout) in CASES:
    gen = torch.Generator(device="cuda").manual_seed(k + n)
    ws = [torch.randn((k, n), generator=gen, device="cuda") for _ in range(4)]
    q1, s1, z1 = ops.rtn_quantize(ws[0], qtype, "group", g, layout=layout)                    # stateful
    q2 = torch.empty_like(q1); s2 = torch.empty_like(s1); z2 = torch.empty_like(z1)
    wsb = torch.empty(lib.oq_rtn_workspace_bytes(k, n, L.OQ_GROUP, g, 0) + 256, dtype=torch.uint8, device="cuda")
    stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)

    def plain(w):
        L.check(lib.oq_rtn_quantize_f32(C.c_void_p(w.data_ptr()), k, n, n, L.QTYPE_CODE[qtype], L.OQ_GROUP, g, 0, 0, 1.0, 0, C.c_void_p(q2.data_ptr()),
                                        C.c_void_p(s2.data_ptr()), C.c_void_p(z2.data_ptr()), ops._layout_code(layout), C.c_void_p(wsb.data_ptr()), wsb.numel(), stream))
    plain(ws[0])
    torch.cuda.synchronize()
    same = bool(torch.equal(q1, q2) and torch.equal(s1.reshape(-1), s2.reshape(-1)) and torch.equal(z1.reshape(-1), z2.reshape(-1)))
    out = (q1, s1.reshape(-1), z1.reshape(-1))
    res = []
    for fn in (lambda w: ops.rtn_quantize(w, qtype, "group", g, layout=layout, out=out), plain):
        for i in range(10):
            fn(ws[i % 4])
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(200):
            fn(ws[i % 4])
        e1.record()
        torch.cuda.synchronize()
        res.append(e0.elapsed_time(e1) * 5.0)
    # the second call on the same state must find it zero: compare again
    plain(ws[0])
    q3, s3, z3 = ops.rtn_quantize(ws[0], qtype, "group", g, layout=layout)
    again = bool(torch.equal(q3, q2) and torch.equal(s3.reshape(-1), s2.reshape(-1)) and torch.equal(z3.reshape(-1), z2.reshape(-1)))
    print(f"{k}x{n} {qtype} g{g} {layout:10s} in-launch {res[0]:7.2f} us   staged + launch {res[1]:7.2f} us   equal {same} / after 200 calls {again}", flush=True)
