"""A GPT-2-architecture export through the writer: projections written as `addmm` on the flattened sequence (transformers' `Conv1D`
-> ONNX Gemm without `transB`), learned position embeddings, LayerNorm, tanh-GELU, causal mask, and an `lm_head` TIED to the token
embedding (the exporter writes `MatMul(x, Transpose(wte))` next to `Gather(wte, ids)`: the weight becomes a constant only once the
Transpose of an initializer is folded, quantize.py:52).

transformers' own GPT-2 does not pass torch's TorchScript exporter offline (`aten::diff` in its mask code), so the model is a plain
torch restatement, exported here by torch's ONNX exporter: another producer's file.
"""
import io
import math
import warnings

import pytest
import torch

from onnx_model_helpers import q_oracle
from onnx_quantize_amd import QActivationArgs, QConfig, QuantType, QWeightArgs
from onnx_quantize_amd import onnx_proto as P
from onnx_quantize_amd.graph_runner import GraphRunner
from onnx_quantize_amd.model_quantize import quantize_model, target_nodes

VOCAB, DIM, HEADS, LAYERS, POSITIONS = 96, 64, 4, 2, 32


class Conv1D(torch.nn.Module):
    def __init__(self, nx, nf):
        super().__init__()
        self.nf = nf
        self.weight = torch.nn.Parameter(torch.randn(nx, nf) * 0.05)
        self.bias = torch.nn.Parameter(torch.randn(nf) * 0.02)

    def forward(self, x):
        return torch.addmm(self.bias, x.view(-1, x.size(-1)), self.weight).view(x.size()[:-1] + (self.nf,))


class Block(torch.nn.Module):
    def __init__(self):
        super().__init__()
        self.ln_1, self.ln_2 = torch.nn.LayerNorm(DIM), torch.nn.LayerNorm(DIM)
        self.c_attn, self.c_proj = Conv1D(DIM, 3 * DIM), Conv1D(DIM, DIM)
        self.c_fc, self.mlp_proj = Conv1D(DIM, 4 * DIM), Conv1D(4 * DIM, DIM)

    def forward(self, x, mask):
        b, t, _ = x.shape
        q, k, v = self.c_attn(self.ln_1(x)).split(DIM, dim=2)
        heads = lambda y: y.view(b, t, HEADS, DIM // HEADS).transpose(1, 2)   # noqa: E731
        a = torch.softmax(heads(q) @ heads(k).transpose(2, 3) / math.sqrt(DIM // HEADS) + mask, dim=-1)
        x = x + self.c_proj((a @ heads(v)).transpose(1, 2).reshape(b, t, DIM))
        return x + self.mlp_proj(torch.nn.functional.gelu(self.c_fc(self.ln_2(x)), approximate="tanh"))


class GPT2(torch.nn.Module):
    def __init__(self):
        super().__init__()
        self.wte, self.wpe = torch.nn.Embedding(VOCAB, DIM), torch.nn.Embedding(POSITIONS, DIM)
        self.h = torch.nn.ModuleList([Block() for _ in range(LAYERS)])
        self.ln_f = torch.nn.LayerNorm(DIM)

    def forward(self, ids):
        t = ids.shape[1]
        x = self.wte(ids) + self.wpe(torch.arange(t, device=ids.device))[None]
        mask = torch.full((t, t), float("-inf"), device=ids.device).triu(1)[None, None]
        for block in self.h:
            x = block(x, mask)
        return torch.nn.functional.linear(self.ln_f(x), self.wte.weight)       # the tied head


@pytest.fixture(scope="module")
def gpt2():
    from torch.onnx._internal.torchscript_exporter import onnx_proto_utils
    onnx_proto_utils._add_onnxscript_fn = lambda proto, _ops: proto          # the exporter's only use of the `onnx` package
    torch.manual_seed(0)
    module = GPT2().eval()
    f = io.BytesIO()
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        torch.onnx.export(module, (torch.randint(0, VOCAB, (2, 10)),), f, dynamo=False, opset_version=17, input_names=["input_ids"], output_names=["logits"],
                          dynamic_axes={"input_ids": {0: "batch", 1: "seq"}, "logits": {0: "batch", 1: "seq"}})
    data = f.getvalue()
    model = P.parse_model(data)
    assert P.serialize(model) == data
    gen = torch.Generator().manual_seed(1)
    return module, model, torch.randint(0, VOCAB, (24, 16), generator=gen).numpy(), torch.randint(0, VOCAB, (8, 16), generator=gen)


def test_the_export_runs_and_its_gemms_and_tied_head_are_targets(gpt2):
    module, model, _calib, _eval = gpt2
    for shape in ((2, 10), (3, 17)):
        ids = torch.randint(0, VOCAB, shape)
        with torch.no_grad():
            want = module(ids)
        torch.testing.assert_close(GraphRunner(model, device="cpu")(ids)["logits"], want, rtol=1e-4, atol=1e-5)
    assert sum(n.op_type == "Gemm" for n in model.graph.node) == 4 * LAYERS
    targets = target_nodes(model, QConfig(weights=QWeightArgs()))
    gemms = [t for t in targets if t[1] == "Gemm"]
    heads = [t for t in targets if t[1] == "MatMul"]
    assert len(gemms) == 4 * LAYERS and sorted(t[3] for t in gemms)[-1] == [4 * DIM, DIM]
    assert len(heads) == 1 and heads[0][3] == [DIM, VOCAB]                      # Transpose(wte) folded: [K, N] = [dim, vocab]
    out = q_oracle(model, QConfig(weights=QWeightArgs(dtype=QuantType.QUInt4, group_size=32)))
    calls = [n for n in out.graph.node if n.domain]
    # gemm_to_qgemm.py:143-147: a Gemm with a constant bias becomes MatMulNBits too, the bias as its sixth input
    assert {n.op_type for n in calls} == {"MatMulNBits"} and len(calls) == 4 * LAYERS + 1
    assert sorted(len([v for v in n.input if v]) for n in calls) == [4] + [5] * (4 * LAYERS) and all(len(n.input) == 6 for n in calls if "Gemm" in n.name)
    signed = q_oracle(model, QConfig(weights=QWeightArgs(dtype=QuantType.QInt4, group_size=32)))          # int4: no MatMulNBits form
    assert sorted({n.op_type for n in signed.graph.node if n.domain}) == ["QGemmWeightsOnlyGrouped", "QMatMulWeightsOnlyGrouped"]
    names = {t.name for t in out.graph.initializer}
    assert "wte.weight" in names                                               # the embedding keeps its float table
    ids = torch.randint(0, VOCAB, (2, 9))
    got, want = GraphRunner(out, device="cpu")(ids)["logits"], GraphRunner(model, device="cpu")(ids)["logits"]
    assert ((got - want).norm() / want.norm()).item() < 0.35
    ignored = q_oracle(model, QConfig(weights=QWeightArgs(dtype=QuantType.QInt8), ignore=["c_attn", "MatMul"]))
    assert sum(bool(n.domain) for n in ignored.graph.node) == 3 * LAYERS


@pytest.mark.gpu
def test_device_files_equal_the_oracle_files_on_the_gpt2_export(gpt2):
    _module, model, calib, evaluation = gpt2
    want = GraphRunner(model, device="cuda")(evaluation)["logits"]
    act = lambda: QActivationArgs(dtype=QuantType.QUInt8, is_static=True)     # noqa: E731
    configs = {
        "uint4_g32": lambda: QConfig(weights=QWeightArgs(dtype=QuantType.QUInt4, group_size=32)),
        "int8_channel": lambda: QConfig(weights=QWeightArgs(dtype=QuantType.QInt8, strategy="channel")),
        "static_in_out": lambda: QConfig(weights=QWeightArgs(dtype=QuantType.QInt8, strategy="channel"), input_activations=act(), output_activations=act(),
                                         calibration_data=calib, calibration_params={"num_samples": 24, "batch_size": 8}),
    }
    for name, make in configs.items():
        out = quantize_model(model, make())
        data = P.serialize(out)
        assert data == P.serialize(q_oracle(model, make(), runner_device="cuda")), name
        got = GraphRunner(P.parse_model(data), device="cuda")(evaluation)["logits"]
        error = ((got - want).norm() / want.norm()).item()
        print(f"{name}: logits rel err {error:.4f}")
        assert error < (0.35 if "uint4" in name else 0.1), (name, error)
