#!/usr/bin/env python3
"""Small ONNX files for the tests of the model reader / writer, written by a producer that is NOT this repository: torch's
TorchScript exporter, whose C++ serialiser is protobuf over the published onnx.proto.  The image has no `onnx` package; the
exporter only needs it for a post-processing hook (custom onnxscript functions), which these models do not use and which is
bypassed here.

    python tests/golden/make_onnx_fixtures.py        # rewrites tests/golden/onnx/*.onnx (seeded: byte-stable per torch build)
"""
import io
import os
import warnings

import torch

HERE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "onnx")


class Block(torch.nn.Module):
    """One pre-norm transformer block at toy width: LayerNorm, q/k/v projections sharing their input, softmax attention,
    output projection, GELU MLP.  Exports to MatMul + Add (3-D inputs), Softmax, Transpose, Reshape, LayerNormalization, Erf."""

    def __init__(self, d=64, heads=4, ff=128):
        super().__init__()
        self.d, self.h = d, heads
        self.ln1, self.ln2 = torch.nn.LayerNorm(d), torch.nn.LayerNorm(d)
        self.q, self.k, self.v = (torch.nn.Linear(d, d, bias=False) for _ in range(3))
        self.o = torch.nn.Linear(d, d)
        self.up, self.down = torch.nn.Linear(d, ff), torch.nn.Linear(ff, d)

    def forward(self, x):
        b, t, d = x.shape
        y = self.ln1(x)
        split = lambda z: z.reshape(b, t, self.h, d // self.h).transpose(1, 2)      # noqa: E731
        q, k, v = split(self.q(y)), split(self.k(y)), split(self.v(y))
        a = torch.softmax(q @ k.transpose(-1, -2) / (d // self.h) ** 0.5, dim=-1)
        x = x + self.o((a @ v).transpose(1, 2).reshape(b, t, d))
        return x + self.down(torch.nn.functional.gelu(self.up(self.ln2(x))))


class Tied(torch.nn.Module):
    """One square matrix read by two MatMuls (one initializer, two consumers: the reference duplicates it before quantizing)."""

    def __init__(self, vocab=40, d=32):
        super().__init__()
        self.emb = torch.nn.Embedding(vocab, d)
        self.w = torch.nn.Parameter(torch.randn(d, d) * 0.2)

    def forward(self, ids):
        return torch.relu(self.emb(ids) @ self.w) @ self.w


def export(model, args, name, **kw):
    from torch.onnx._internal.torchscript_exporter import onnx_proto_utils
    onnx_proto_utils._add_onnxscript_fn = lambda proto, _ops: proto
    f = io.BytesIO()
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        torch.onnx.export(model.eval(), args, f, dynamo=False, opset_version=kw.pop("opset", 17), **kw)
    path = os.path.join(HERE, name)
    with open(path, "wb") as out:
        out.write(f.getvalue())
    print(f"{name}: {len(f.getvalue())} bytes")


def main():
    os.makedirs(HERE, exist_ok=True)
    torch.manual_seed(0)
    mlp = torch.nn.Sequential(torch.nn.Linear(64, 48), torch.nn.ReLU(), torch.nn.Linear(48, 16, bias=False))
    # 2-D input: Linear exports as Gemm with transB = 1 (the standardise-Gemm pre-pass has something to do)
    export(mlp, (torch.randn(4, 64),), "mlp_gemm.onnx", input_names=["x"], output_names=["y"],
           dynamic_axes={"x": {0: "batch"}, "y": {0: "batch"}})
    # 3-D input: Linear exports as MatMul (+ Add)
    export(mlp, (torch.randn(2, 5, 64),), "mlp_matmul.onnx", input_names=["x"], output_names=["y"],
           dynamic_axes={"x": {0: "batch", 1: "seq"}, "y": {0: "batch", 1: "seq"}})
    torch.manual_seed(1)
    export(Block(), (torch.randn(2, 6, 64),), "block.onnx", input_names=["x"], output_names=["y"],
           dynamic_axes={"x": {0: "batch"}, "y": {0: "batch"}})
    torch.manual_seed(2)
    export(Tied(), (torch.randint(0, 40, (2, 7)),), "tied.onnx", input_names=["ids"], output_names=["logits"])
    torch.manual_seed(3)
    wide = torch.nn.Sequential(torch.nn.Linear(256, 128, bias=False), torch.nn.Tanh(), torch.nn.Linear(128, 256, bias=False))
    export(wide, (torch.randn(2, 3, 256),), "wide_matmul.onnx", input_names=["x"], output_names=["y"], opset=21,
           dynamic_axes={"x": {0: "batch"}, "y": {0: "batch"}})


if __name__ == "__main__":
    main()
