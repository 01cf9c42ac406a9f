#!/usr/bin/env python3
"""The two launches whose `roofline.traffic` had no counter record, alone in one process so that a
rocprofv3 pass over it (`--kernel-trace --stats`, `--pmc FETCH_SIZE`, `--pmc WRITE_SIZE`, each in its own run, the program directly
after `--`) sees nothing else:
  calibration  oq::minmax_partial<float> on the 64 x 2048 x 2560 fp32 tensor of bench_calib.py's `roofline` (1.34 GB), 20 launches
  hessian      ops.hessian_accumulate (method auto = fp16 pieces) on bench_gptq.py's widest batch, 65 536 x 11008 fp32, 4 calls
usage: pmc_targets.py calibration|hessian"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from onnx_quantize_amd.hip import ops  # noqa: E402

what = sys.argv[1]
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
gen = torch.Generator(device=dev).manual_seed(3)
if what == "calibration":
    big = torch.randn((64, 2048, 2560), generator=gen, device=dev)
    st = ops.minmax_state(dev)
    for _ in range(20):
        ops.minmax_collect(big, st)
    torch.cuda.synchronize()
    print("calibration: 20 launches over", big.numel() * 4, "bytes")
else:
    x = torch.randn((32, 2048, 11008), generator=gen, device=dev)
    h = torch.zeros((11008, 11008), device=dev)
    n = 0
    for _ in range(4):
        n = ops.hessian_accumulate(x, h, n)
    torch.cuda.synchronize()
    print("hessian: 4 calls of", x.shape[0] * x.shape[1], "x", x.shape[2])
