#!/usr/bin/env python3
"""Per-tensor / per-channel / tall-group RTN on the Llama shapes: time per call (HIP events on the launch stream, rotating
inputs so that reads come from HBM), fraction of the 8 TB/s roofline in algorithmic bytes, digest of the outputs.
OQ_RTN_RESIDENT=0 selects the three-launch path that reads W twice (rounds 1-3); the digests of the two must agree.

    python scripts/quick_strategies.py [--reps 200] [--json out.json]
"""
import argparse
import hashlib
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if "--lib" in sys.argv:      # lab: another build of the library (same ABI), e.g. the previous commit's, for a same-box comparison
    from onnx_quantize_amd.hip import _lib  # noqa: E402
    _lib.LIB_PATH = os.path.abspath(sys.argv[sys.argv.index("--lib") + 1])
from onnx_quantize_amd.hip import ops  # noqa: E402


def sha(t):
    return hashlib.sha256(t.contiguous().cpu().numpy().tobytes()).hexdigest()[:16]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=200)
    ap.add_argument("--lib", default=None, help="lab: path of another build of liboq_hip.so")
    ap.add_argument("--json", default=None)
    ap.add_argument("--shapes", default="4096x11008,4096x4096,11008x4096,256x512,640x2048")
    ap.add_argument("--only", default=None, help="one strategy only: channel | tensor | group")
    args = ap.parse_args()
    torch.cuda.set_device(0)
    out = {"resident": os.environ.get("OQ_RTN_RESIDENT", "1"), "rows": []}
    for shp in args.shapes.split(","):
        k, n = (int(v) for v in shp.split("x"))
        rot = max(2, min(6, (800 << 20) // (k * n * 4)))
        gen = torch.Generator(device="cuda").manual_seed(k + n)
        ws = [torch.randn((k, n), generator=gen, device="cuda") for _ in range(rot)]
        for qtype, strategy, g in (("int8", "channel", -1), ("int8", "tensor", -1), ("uint4", "channel", -1), ("int8", "group", 512),
                                   ("int8", "group", 1024)):
            if (strategy == "group" and k % g) or (args.only and strategy != args.only):
                continue
            outs = ops.rtn_quantize(ws[0], qtype, strategy, g)
            dig = "/".join(sha(t) for t in outs)
            torch.cuda.synchronize()
            for i in range(10):
                ops.rtn_quantize(ws[i % rot], qtype, strategy, g, out=outs)
            st, en = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            st.record()
            for i in range(args.reps):
                ops.rtn_quantize(ws[i % rot], qtype, strategy, g, out=outs)
            en.record()
            torch.cuda.synchronize()
            us = st.elapsed_time(en) * 1e3 / args.reps
            nparam = (1 if strategy == "tensor" else n * (k // (k if g == -1 else g)))
            alg = k * n * 4 + k * n + nparam * 5                  # W once, one byte per value (the [K, N] container), (scale, zp)
            row = dict(shape=shp, qtype=qtype, strategy=strategy, g=g, us=round(us, 2), alg_bytes=alg, tbps=round(alg / us / 1e6, 3),
                       frac=round(alg / us / 1e6 / 8.0, 3), digest=dig)
            out["rows"].append(row)
            print(json.dumps(row), flush=True)
        del ws
        torch.cuda.empty_cache()
    try:        # a -DOQ_SPIN_LIMIT lab build counts the spins that gave up (results are garbage then)
        import ctypes
        from onnx_quantize_amd.hip import _lib as L_
        n = ctypes.c_uint32(0)
        if ctypes.CDLL(L_.LIB_PATH).oq_lab_spin_timeouts(ctypes.byref(n), 0) == 0:
            out["spin_timeouts"] = n.value
            print(json.dumps({"spin_timeouts": n.value}), flush=True)
    except AttributeError:
        pass
    if args.json:
        with open(args.json, "w") as f:
            json.dump(out, f, indent=1)


if __name__ == "__main__":
    main()
