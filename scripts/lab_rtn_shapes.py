#!/usr/bin/env python3
"""Group RTN on several shapes under ONE setting of the OQ_RTN_* knobs (they are read once per process): time per launch
through the C ABI (HIP events on the launch stream, inputs rotating over > 256 MiB), fraction of the 8 TB/s roofline in
algorithmic bytes, digest of the outputs (settings must agree on it).

    OQ_RTN_XG=2 python scripts/lab_rtn_shapes.py --shapes 4096x4096,11008x4096 --layout nbits
"""
import argparse
import ctypes as C
import hashlib
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from onnx_quantize_amd.hip import _lib as L  # noqa: E402


def sha(*ts):
    h = hashlib.sha256()
    for t in ts:
        h.update(t.contiguous().cpu().numpy().tobytes())
    return h.hexdigest()[:12]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--shapes", default="4096x11008,4096x4096,11008x4096,8192x8192")
    ap.add_argument("--layout", default="nbits", choices=["nbits", "kn", "kn_packed4"])
    ap.add_argument("--qtype", default="uint4")
    ap.add_argument("--g", type=int, default=128)
    ap.add_argument("--reps", type=int, default=300)
    ap.add_argument("--trials", type=int, default=3)
    ap.add_argument("--lib", default=None, help="lab: another build of liboq_hip.so (scripts/lab_build_variant.sh)")
    args = ap.parse_args()
    if args.lib:
        L.LIB_PATH = os.path.abspath(args.lib)
    lib = L.load()
    torch.cuda.set_device(0)
    stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    layout = {"nbits": L.OQ_LAYOUT_NBITS, "kn": L.OQ_LAYOUT_KN, "kn_packed4": L.OQ_LAYOUT_KN_PACKED4}[args.layout]
    bits = 4 if "4" in args.qtype else 8
    knobs = {k: v for k, v in os.environ.items() if k.startswith("OQ_RTN_")}
    for shp in args.shapes.split(","):
        k, n = (int(v) for v in shp.split("x"))
        rot = max(3, min(12, (900 << 20) // (k * n * 4)))
        gen = torch.Generator(device="cuda").manual_seed(k * 7 + n)
        ws = [torch.randn((k, n), generator=gen, device="cuda") for _ in range(rot)]
        groups = k * n // args.g
        qbytes = k * n if args.layout == "kn" else k * n * bits // 8
        outs = [(torch.empty(qbytes, dtype=torch.uint8, device="cuda"), torch.empty(groups, dtype=torch.float32, device="cuda"),
                 torch.empty(groups, dtype=torch.uint8, device="cuda")) for _ in range(4)]
        wsb = lib.oq_rtn_workspace_bytes(k, n, L.OQ_GROUP, args.g, 0)
        wsbuf = torch.empty(max(wsb, 256), dtype=torch.uint8, device="cuda")
        calls = [(C.c_void_p(w.data_ptr()),) for w in ws]
        optr = [(C.c_void_p(q.data_ptr()), C.c_void_p(s.data_ptr()), C.c_void_p(z.data_ptr())) for q, s, z in outs]
        wsp, wsn = C.c_void_p(wsbuf.data_ptr()), wsbuf.numel()
        qt = L.QTYPE_CODE[args.qtype]

        # the stateful entry point with a zeroed, self-cleaning state: what ops.rtn_quantize calls ([K,N] layouts: parameters transposed
        # inside the launch since round 6); OQ_LAB_NO_STATE=1: the plain entry point (staged parameters + a transpose launch)
        nstate = 0 if os.environ.get("OQ_LAB_NO_STATE") == "1" else lib.oq_rtn_state_bytes(k, n, L.OQ_GROUP, args.g)
        state = torch.zeros(nstate + 256, dtype=torch.uint8, device="cuda")
        stp, stn = (C.c_void_p(state.data_ptr()), state.numel()) if nstate else (C.c_void_p(0), 0)

        def step(i):
            qp, sp, zp = optr[i % 4]
            st = lib.oq_rtn_quantize_stateful_f32(calls[i % rot][0], k, n, n, qt, L.OQ_GROUP, args.g, 0, 0, 1.0, 0, qp, sp, zp, layout, wsp, wsn, stp, stn, stream)
            if st != 0:
                L.check(st)
        step(0)
        torch.cuda.synchronize()
        dig = sha(*outs[0])
        times = []
        for _ in range(args.trials):
            for i in range(20):
                step(i)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for i in range(args.reps):
                step(i)
            e1.record()
            torch.cuda.synchronize()
            times.append(round(e0.elapsed_time(e1) * 1e3 / args.reps, 2))
        alg = k * n * 4 + k * n * bits // 8 + groups * 5
        print(json.dumps(dict(shape=shp, layout=args.layout, knobs=knobs, us=times, frac=round(alg / min(times) / 1e6 / 8.0, 4), digest=dig)), flush=True)
        del ws, outs, wsbuf
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
