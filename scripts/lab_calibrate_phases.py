#!/usr/bin/env python3
"""Inside one calibration walk of the 7B-width decoder model of bench_model.py: runner construction (weights to HBM), the graph
passes, the stream's consumers (Hessian updates), per batch.

    python scripts/lab_calibrate_phases.py [--layers 8] [--samples 32] [--seq 2048]
"""
import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench_model  # noqa: E402
from onnx_quantize_amd.calibration_driver import ActivationStream  # noqa: E402
from onnx_quantize_amd.graph_runner import GraphRunner  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--layers", type=int, default=8)
    ap.add_argument("--samples", type=int, default=32)
    ap.add_argument("--seq", type=int, default=2048)
    ap.add_argument("--batch", type=int, default=8)
    a = ap.parse_args()
    model = bench_model.build_model(a.layers, 4096, 11008)
    targets = [n for n in model.graph.node if n.op_type == "MatMul"]
    wanted = list(dict.fromkeys(n.input[0] for n in targets))
    torch.zeros(1, device="cuda")

    def clock():
        torch.cuda.synchronize()
        return time.perf_counter()
    t0 = clock()
    runner = GraphRunner(model, outputs=wanted, device="cuda", capture=True)
    t_build = clock() - t0
    stream = ActivationStream(hessian_names=wanted)
    data = torch.randn(a.samples, a.seq, 4096, generator=torch.Generator().manual_seed(1))
    rows = []
    for i in range(0, a.samples, a.batch):
        t0 = clock()
        x = data[i:i + a.batch].cuda()
        t1 = clock()
        acts = runner(x)
        t2 = clock()
        stream.feed(acts)
        t3 = clock()
        del acts
        rows.append({"h2d_ms": round((t1 - t0) * 1e3, 1), "pass_ms": round((t2 - t1) * 1e3, 1), "feed_ms": round((t3 - t2) * 1e3, 1),
                     "allocated_gb": round(torch.cuda.max_memory_allocated() / 1e9, 1)})
    print(json.dumps({"layers": a.layers, "runner_build_s": round(t_build, 3), "tapped_values": len(wanted), "batches": rows}))


if __name__ == "__main__":
    main()
