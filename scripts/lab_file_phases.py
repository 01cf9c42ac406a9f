#!/usr/bin/env python3
"""Where the time of a calibrated `quantize_model` run goes: wall time per phase (calibration walks, searches, emission) on the
gemma-3-270m-sized genai-style file of examples/gemma3_shapes/gemma3_onnx_file.py with the reference's AWQ example configuration.

    python scripts/lab_file_phases.py [--layers 18] [--samples 64] [--block 256]
"""
import argparse
import importlib.util
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import onnx_quantize_amd.model_quantize as MQ  # noqa: E402
from onnx_quantize_amd import AwqConfig, CalibrationParams, QConfig, QWeightArgs  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--layers", type=int, default=18)
    ap.add_argument("--vocab", type=int, default=32768)
    ap.add_argument("--samples", type=int, default=64)
    ap.add_argument("--block", type=int, default=256)
    a = ap.parse_args()
    spec = importlib.util.spec_from_file_location("g", os.path.join(ROOT, "examples", "gemma3_shapes", "gemma3_onnx_file.py"))
    ex = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ex)
    model = ex.build_model(a.layers, a.vocab)
    torch.zeros(1, device="cuda")
    phases = {}

    def timed(name, fn):
        def wrapper(*args, **kw):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            out = fn(*args, **kw)
            torch.cuda.synchronize()
            phases.setdefault(name, []).append(round(time.perf_counter() - t0, 4))
            return out
        return wrapper

    MQ._calibrate = timed("calibrate", MQ._calibrate)
    MQ._preprocess = timed("preprocess", MQ._preprocess)
    MQ.plan_node = timed("plan_node", MQ.plan_node)
    for rep in range(2):
        phases.clear()
        qc = QConfig(weights=QWeightArgs(dtype="uint4", strategy="group", group_size=128), preprocessors=[AwqConfig()],
                     calibration_data=ex.make_calibration_data(a.layers, a.vocab, a.samples, a.block),
                     calibration_params=CalibrationParams(batch_size=1, num_samples=a.samples), ignore=["lm_head"])
        t0 = time.perf_counter()
        MQ.quantize_model(model, qc)
        torch.cuda.synchronize()
        total = time.perf_counter() - t0
        print(json.dumps({"rep": rep, "total_s": round(total, 3), "calibrate_s": phases.get("calibrate"), "preprocess_s": phases.get("preprocess"),
                          "plan_nodes_s": round(sum(phases.get("plan_node", [])), 4), "nodes": len(phases.get("plan_node", []))}), flush=True)


if __name__ == "__main__":
    main()
