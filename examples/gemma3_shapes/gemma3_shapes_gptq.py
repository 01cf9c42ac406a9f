"""Device-resident counterpart of the reference's `examples/gemma3/` scripts, on a gemma-3-270m-SHAPED torch model (random
weights: the real checkpoint, wikitext and onnxruntime-genai are not available offline).

What the reference does per model (`quantize()` -> pre-passes -> `calibrate_model` -> rewrite rules, all on NumPy arrays):
  1. run the float model on the calibration batches and keep every tapped activation on the host (calibrate.py:204-251),
  2. static activation ranges from those lists (calibrate.py:254-285), GPTQ Hessians from their concatenation (:292-307),
  3. quantize every MatMul weight (qrules/_common.py:126-142) and emit MatMulNBits initializers (:65-123).
Here the same three steps never leave the GPU: `TorchRunner` taps the Linear inputs / outputs, `ActivationStream.feed` folds a
batch into min-max state and the running Hessians (one grouped launch chain per batch), `quantize_weights_gptq` factors every
distinct input once and runs the GPTQ loop, `wire_format` packs the MatMulNBits blobs.

    python examples/gemma3_shapes/gemma3_shapes_gptq.py [--layers 18] [--batches 51] [--mode parity|corrected]
"""
from __future__ import annotations

import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))


def build_model(layers: int, dev):
    import torch

    class Block(torch.nn.Module):          # gemma-3-270m's MatMul shapes: hidden 640, q/k/v 1024 wide, MLP 2048
        def __init__(self):
            super().__init__()
            self.qkv = torch.nn.Linear(640, 1024, bias=False)
            self.o = torch.nn.Linear(1024, 640, bias=False)
            self.up = torch.nn.Linear(640, 2048, bias=False)
            self.down = torch.nn.Linear(2048, 640, bias=False)

        def forward(self, x):
            x = x + self.o(torch.tanh(self.qkv(x)))
            return x + self.down(torch.nn.functional.gelu(self.up(x)))

    torch.manual_seed(0)
    return torch.nn.Sequential(*[Block() for _ in range(layers)]).to(dev)


def main(layers: int = 18, batches: int = 51, mode: str = "parity", verbose: bool = True) -> dict:
    import torch

    from onnx_quantize_amd.calibration import MinMaxCalibrator
    from onnx_quantize_amd.calibration_driver import ActivationStream, TorchRunner, quantize_weights_gptq
    from onnx_quantize_amd.config import QActivationArgs
    from onnx_quantize_amd.dtypes import QuantType
    from onnx_quantize_amd.hip import ops

    dev = torch.device("cuda", 0)
    model = build_model(layers, dev)
    names = ("qkv", "o", "up", "down")
    taps, in_names, out_names = {}, [], []
    for i in range(layers):
        for n in names:
            taps[f"{i}.{n}/in"] = (f"{i}.{n}", "input")
            taps[f"{i}.{n}/out"] = (f"{i}.{n}", "output")
            in_names.append(f"{i}.{n}/in")
            out_names.append(f"{i}.{n}/out")
    runner = TorchRunner(model, taps)
    stream = ActivationStream(calibrator=MinMaxCalibrator(), input_names=in_names, output_names=out_names, hessian_names=in_names)
    g = torch.Generator(device=dev).manual_seed(1)
    feeds = [torch.randn((10, 512, 640), generator=g, device=dev) for _ in range(4)]      # reused round robin
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    with torch.no_grad():
        for b in range(batches):
            stream.feed(runner(feeds[b % len(feeds)]))           # step 1 + 2, batch by batch in HBM
    args = QActivationArgs(dtype=QuantType.QUInt8, is_static=True)
    act_in, act_out = stream.input_qparams(args), stream.output_qparams(args)
    torch.cuda.synchronize()
    t_cal = time.perf_counter() - t0

    weights = {f"{i}.{n}": (getattr(model[i], n).weight.detach().t().contiguous(), f"{i}.{n}/in") for i in range(layers) for n in names}
    t1 = time.perf_counter()
    quantized = quantize_weights_gptq(weights, stream.hessians, "uint4", "group", 128, mode=mode)                 # step 3
    blobs = {k: (ops.pack_matmul_nbits(q, 128, 4), s, z) for k, (q, s, z, _) in quantized.items()}                 # MatMulNBits B
    torch.cuda.synchronize()
    t_q = time.perf_counter() - t1
    params = sum(w.numel() for w, _ in weights.values())
    out = {"layers": layers, "batches": batches, "mode": mode, "weights": len(weights), "params": params,
           "activation_qparams": len(act_in) + len(act_out), "calibration_s": round(t_cal, 3), "gptq_and_packing_s": round(t_q, 3),
           "blob_shape_of_0.qkv": tuple(blobs["0.qkv"][0].shape)}
    if verbose:
        print(out)
    return out, blobs, quantized, weights


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--layers", type=int, default=18)
    ap.add_argument("--batches", type=int, default=51)
    ap.add_argument("--mode", choices=["parity", "corrected"], default="parity")
    a = ap.parse_args()
    main(a.layers, a.batches, a.mode)
