#!/usr/bin/env python3
"""First contact of the one-read channel / tensor kernels (rtn_resident.hip) with the GPU: small shapes against the oracle,
then the Llama shapes for self-consistency (every integer within its range's parameters).  Run under a short `timeout`."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import oq_oracle as O  # noqa: E402
from onnx_quantize_amd.hip import ops  # noqa: E402

torch.cuda.set_device(0)
for (k, n) in ((256, 512), (1536, 516), (130, 2052), (2048, 4096)):
    w = np.random.default_rng(k + n).standard_t(3, size=(k, n)).astype(np.float32)
    wd = torch.from_numpy(w).cuda()
    for qtype, strategy, g, sym in (("int8", "tensor", -1, True), ("int8", "channel", -1, False), ("uint4", "channel", -1, False),
                                    ("uint8", "tensor", -1, False), ("int8", "group", 512, False), ("int4", "group", 384, True)):
        if strategy == "group" and k % g:
            continue
        q, s, z = ops.rtn_quantize(wd, qtype, strategy, g, sym, False, 0.95)
        torch.cuda.synchronize()
        eq, es, ez = O.rtn_quantize(w, qtype, strategy, g, sym, False, 0.95)
        ok = np.array_equal(q.cpu().numpy(), eq) and s.cpu().numpy().tobytes() == np.asarray(es).tobytes() and np.array_equal(z.cpu().numpy(), ez)
        print(k, n, qtype, strategy, g, "OK" if ok else "MISMATCH", flush=True)
        if not ok:
            gs, gz, gq = s.cpu().numpy().reshape(-1), z.cpu().numpy().reshape(-1), q.cpu().numpy()
            es1, ez1 = np.asarray(es).reshape(-1), np.asarray(ez).reshape(-1)
            bad_s = np.nonzero(gs.view(np.uint32) != es1.view(np.uint32))[0]
            bad_z = np.nonzero(gz != ez1)[0]
            bad_q = np.argwhere(gq != eq)
            print("  scale mismatches", bad_s.size, bad_s[:12], gs[bad_s[:6]], es1[bad_s[:6]])
            print("  zp mismatches", bad_z.size, bad_z[:12], gz[bad_z[:6]], ez1[bad_z[:6]])
            print("  q mismatches", len(bad_q), bad_q[:8].tolist())
            if len(bad_q):
                rows = np.unique(bad_q[:, 0]); cols = np.unique(bad_q[:, 1])
                print("  rows", rows[:20], "...", rows[-5:], "cols", cols[:20], "...", cols[-5:])
            sys.exit(1)
for (k, n) in ((4096, 4096), (4096, 11008), (11008, 4096)):
    wd = torch.randn((k, n), device="cuda")
    for strategy in ("tensor", "channel"):
        q, s, z = ops.rtn_quantize(wd, "int8", strategy, -1)
        torch.cuda.synchronize()
        dq = (q.float() - z.float()) * s
        err = float((dq - wd).abs().max() / s.max())
        print(k, n, strategy, "max |w - dq| / scale =", round(err, 4), flush=True)
        assert err <= 0.5001
print("sanity ok")
