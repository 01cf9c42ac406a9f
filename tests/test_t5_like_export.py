"""A T5-architecture encoder-decoder export through the writer: bucketed relative position bias (Abs / Log / Min / Where / Less /
Gather on an embedding of buckets), RMS norm without bias, bias-free projections, cross attention over the encoder's output, ReLU
feed-forward, a tied and rescaled head -- two model inputs (`input_ids`, `decoder_input_ids`), MatMuls with shared inputs on both
sides of the cross attention.

A plain torch restatement of the architecture, exported here by torch's ONNX exporter (another producer's file)."""
import io
import math
import warnings

import numpy as np
import pytest
import torch

from onnx_model_helpers import q_oracle
from onnx_quantize_amd import QActivationArgs, QConfig, QuantType, QWeightArgs
from onnx_quantize_amd import onnx_proto as P
from onnx_quantize_amd.graph_runner import GraphRunner
from onnx_quantize_amd.model_quantize import quantize_model, target_nodes

VOCAB, DIM, HEADS, FF, BUCKETS, MAX_DISTANCE = 80, 64, 4, 128, 8, 20


class Norm(torch.nn.Module):
    def __init__(self):
        super().__init__()
        self.weight = torch.nn.Parameter(1 + 0.1 * torch.randn(DIM))

    def forward(self, x):
        return self.weight * x * torch.rsqrt(x.pow(2).mean(-1, keepdim=True) + 1e-6)


def bucket(relative, bidirectional):
    n = BUCKETS
    ret = torch.zeros_like(relative)
    if bidirectional:
        n = n // 2
        ret = ret + (relative > 0).long() * n
        relative = relative.abs()
    else:
        relative = -torch.min(relative, torch.zeros_like(relative))
    exact = n // 2
    large = exact + (torch.log(relative.float() / exact) / math.log(MAX_DISTANCE / exact) * (n - exact)).long()
    large = torch.min(large, torch.full_like(large, n - 1))
    return ret + torch.where(relative < exact, relative, large)


class Attention(torch.nn.Module):
    def __init__(self, bias_table=False):
        super().__init__()
        lin = lambda: torch.nn.Linear(DIM, DIM, bias=False)                     # noqa: E731
        self.q, self.k, self.v, self.o = lin(), lin(), lin(), lin()
        self.table = torch.nn.Embedding(BUCKETS, HEADS) if bias_table else None

    def forward(self, x, memory, bias):
        b, t, s = x.shape[0], x.shape[1], memory.shape[1]
        heads = lambda y, n: y.view(b, n, HEADS, DIM // HEADS).transpose(1, 2)  # noqa: E731
        scores = heads(self.q(x), t) @ heads(self.k(memory), s).transpose(2, 3)  # (T5 does not scale the scores)
        return self.o((torch.softmax(scores + bias, dim=-1) @ heads(self.v(memory), s)).transpose(1, 2).reshape(b, t, DIM))


class T5(torch.nn.Module):
    def __init__(self):
        super().__init__()
        self.shared = torch.nn.Embedding(VOCAB, DIM)
        self.enc_attn, self.dec_attn, self.cross = Attention(True), Attention(True), Attention()
        self.norms = torch.nn.ModuleList([Norm() for _ in range(7)])
        self.wi = torch.nn.ModuleList([torch.nn.Linear(DIM, FF, bias=False) for _ in range(2)])
        self.wo = torch.nn.ModuleList([torch.nn.Linear(FF, DIM, bias=False) for _ in range(2)])

    def position_bias(self, attn, t, s, bidirectional, device):
        relative = torch.arange(s, device=device)[None, :] - torch.arange(t, device=device)[:, None]
        bias = attn.table(bucket(relative, bidirectional)).permute(2, 0, 1)[None]
        if not bidirectional:
            bias = bias + torch.full((t, s), float("-inf"), device=device).triu(1)[None, None]
        return bias

    def forward(self, ids, decoder_ids):
        n = self.norms
        x = self.shared(ids)
        y = n[0](x)
        x = x + self.enc_attn(y, y, self.position_bias(self.enc_attn, ids.shape[1], ids.shape[1], True, ids.device))
        x = x + self.wo[0](torch.relu(self.wi[0](n[1](x))))
        memory = n[2](x)
        d = self.shared(decoder_ids)
        y = n[3](d)
        d = d + self.dec_attn(y, y, self.position_bias(self.dec_attn, decoder_ids.shape[1], decoder_ids.shape[1], False, ids.device))
        d = d + self.cross(n[4](d), memory, torch.zeros(1, 1, 1, 1, device=ids.device))
        d = d + self.wo[1](torch.relu(self.wi[1](n[5](d))))
        return torch.nn.functional.linear(n[6](d) * DIM ** -0.5, self.shared.weight)


@pytest.fixture(scope="module")
def t5():
    from torch.onnx._internal.torchscript_exporter import onnx_proto_utils
    onnx_proto_utils._add_onnxscript_fn = lambda proto, _ops: proto
    torch.manual_seed(0)
    module = T5().eval()
    f = io.BytesIO()
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        torch.onnx.export(module, (torch.randint(0, VOCAB, (2, 9)), torch.randint(0, VOCAB, (2, 6))), f, dynamo=False, opset_version=17,
                          input_names=["input_ids", "decoder_input_ids"], output_names=["logits"],
                          dynamic_axes={"input_ids": {0: "batch", 1: "src"}, "decoder_input_ids": {0: "batch", 1: "tgt"}, "logits": {0: "batch", 1: "tgt"}})
    data = f.getvalue()
    model = P.parse_model(data)
    assert P.serialize(model) == data
    gen = torch.Generator().manual_seed(1)
    feed = lambda n: {"input_ids": torch.randint(0, VOCAB, (n, 14), generator=gen).numpy(),     # noqa: E731
                      "decoder_input_ids": torch.randint(0, VOCAB, (n, 11), generator=gen).numpy()}
    return module, model, feed(24), feed(8)


def test_the_export_runs_to_what_the_module_computes(t5):
    module, model, _calib, _eval = t5
    for src, tgt in ((9, 6), (13, 17), (30, 3)):
        a, b = torch.randint(0, VOCAB, (3, src)), torch.randint(0, VOCAB, (3, tgt))
        with torch.no_grad():
            want = module(a, b)
        got = GraphRunner(model, device="cpu")({"input_ids": a, "decoder_input_ids": b})["logits"]
        torch.testing.assert_close(got, want, rtol=1e-4, atol=1e-5)
    targets = target_nodes(model, QConfig(weights=QWeightArgs()))
    assert len(targets) == 12 + 4 + 1 and all(t[1] == "MatMul" for t in targets)      # three attentions, two feed-forwards, the tied head
    out = q_oracle(model, QConfig(weights=QWeightArgs(dtype=QuantType.QUInt4, group_size=32)))
    assert sum(n.op_type == "MatMulNBits" for n in out.graph.node) == 17


@pytest.mark.gpu
def test_device_files_equal_the_oracle_files_on_the_t5_export(t5):
    _module, model, calib, evaluation = t5
    feed = {k: torch.from_numpy(v) for k, v in evaluation.items()}
    want = GraphRunner(model, device="cuda")(feed)["logits"]
    act = lambda: QActivationArgs(dtype=QuantType.QUInt8, is_static=True)      # noqa: E731
    configs = {
        "uint4_g32": lambda: QConfig(weights=QWeightArgs(dtype=QuantType.QUInt4, group_size=32)),
        "static_in_out": lambda: QConfig(weights=QWeightArgs(dtype=QuantType.QInt8, strategy="channel"), input_activations=act(), output_activations=act(),
                                         calibration_data=calib, calibration_params={"num_samples": 24, "batch_size": 8}),
    }
    for name, make in configs.items():
        data = P.serialize(quantize_model(model, make()))
        assert data == P.serialize(q_oracle(model, make(), runner_device="cuda")), name
        got = GraphRunner(P.parse_model(data), device="cuda")(feed)["logits"]
        error = ((got - want).norm() / want.norm()).item()
        print(f"{name}: logits rel err {error:.4f}")
        assert np.isfinite(error) and error < 0.5, (name, error)
