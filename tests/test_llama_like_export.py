"""A Llama-architecture export (RMSNorm as Pow / ReduceMean / Sqrt, rotary embedding as Cos / Sin / Slice / Neg / Concat, grouped
key / value heads, SwiGLU, causal mask) through the writer: the architecture BASELINE.json's GPTQ configurations are about.

The model is a plain torch restatement of the Llama block (transformers' own class does not pass torch's TorchScript exporter
offline) exported here by torch's ONNX exporter, so the file is another producer's: it must round-trip byte for byte, run in
`GraphRunner` to what the module computes, and quantize under BASELINE configurations 2 and 4 (uint4 groups -> MatMulNBits; GPTQ
int4 groups with calibration through the graph), `lm_head` ignored as in the reference's LLM examples.
"""
import io
import math
import warnings

import numpy as np
import pytest
import torch

from onnx_model_helpers import q_oracle
from onnx_quantize_amd import GPTQConfig, QConfig, QuantType, QWeightArgs
from onnx_quantize_amd import onnx_proto as P
from onnx_quantize_amd.graph_runner import GraphRunner
from onnx_quantize_amd.model_quantize import quantize_model

VOCAB, DIM, HEADS, KV, FFN, LAYERS = 96, 64, 4, 2, 128, 2


class RMSNorm(torch.nn.Module):
    def __init__(self, d):
        super().__init__()
        self.weight = torch.nn.Parameter(torch.ones(d) + 0.1 * torch.randn(d))

    def forward(self, x):
        return self.weight * (x * torch.rsqrt(x.pow(2).mean(-1, keepdim=True) + 1e-6))


def rotate_half(x):
    half = x.shape[-1] // 2
    return torch.cat((-x[..., half:], x[..., :half]), dim=-1)


class Layer(torch.nn.Module):
    def __init__(self):
        super().__init__()
        hd = DIM // HEADS
        lin = lambda i, o: torch.nn.Linear(i, o, bias=False)                  # noqa: E731
        self.q_proj, self.k_proj, self.v_proj, self.o_proj = lin(DIM, HEADS * hd), lin(DIM, KV * hd), lin(DIM, KV * hd), lin(HEADS * hd, DIM)
        self.gate_proj, self.up_proj, self.down_proj = lin(DIM, FFN), lin(DIM, FFN), lin(FFN, DIM)
        self.input_layernorm, self.post_attention_layernorm = RMSNorm(DIM), RMSNorm(DIM)

    def forward(self, x, cos, sin, mask):
        b, t, _ = x.shape
        hd, rep = DIM // HEADS, HEADS // KV
        y = self.input_layernorm(x)
        q = self.q_proj(y).view(b, t, HEADS, hd).transpose(1, 2)
        k = self.k_proj(y).view(b, t, KV, hd).transpose(1, 2)
        v = self.v_proj(y).view(b, t, KV, hd).transpose(1, 2)
        q, k = q * cos + rotate_half(q) * sin, k * cos + rotate_half(k) * sin
        k = k[:, :, None].expand(b, KV, rep, t, hd).reshape(b, HEADS, t, hd)
        v = v[:, :, None].expand(b, KV, rep, t, hd).reshape(b, HEADS, t, hd)
        a = torch.softmax(q @ k.transpose(2, 3) / math.sqrt(hd) + mask, dim=-1)
        x = x + self.o_proj((a @ v).transpose(1, 2).reshape(b, t, HEADS * hd))
        y = self.post_attention_layernorm(x)
        return x + self.down_proj(torch.nn.functional.silu(self.gate_proj(y)) * self.up_proj(y))


class Llama(torch.nn.Module):
    def __init__(self):
        super().__init__()
        self.embed_tokens = torch.nn.Embedding(VOCAB, DIM)
        self.layers = torch.nn.ModuleList([Layer() for _ in range(LAYERS)])
        self.norm, self.lm_head = RMSNorm(DIM), torch.nn.Linear(DIM, VOCAB, bias=False)
        hd = DIM // HEADS
        self.register_buffer("inv_freq", 1.0 / (10000 ** (torch.arange(0, hd, 2).float() / hd)), persistent=False)

    def forward(self, ids):
        t = ids.shape[1]
        x = self.embed_tokens(ids)
        freqs = torch.outer(torch.arange(t, device=ids.device).float(), self.inv_freq)
        emb = torch.cat((freqs, freqs), -1)
        cos, sin = emb.cos()[None, None], emb.sin()[None, None]
        mask = torch.full((t, t), float("-inf"), device=ids.device).triu(1)[None, None]
        for layer in self.layers:
            x = layer(x, cos, sin, mask)
        return self.lm_head(self.norm(x))


@pytest.fixture(scope="module")
def llama():
    from torch.onnx._internal.torchscript_exporter import onnx_proto_utils
    onnx_proto_utils._add_onnxscript_fn = lambda proto, _ops: proto          # the exporter's only use of the `onnx` package
    torch.manual_seed(0)
    module = Llama().eval()
    f = io.BytesIO()
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        torch.onnx.export(module, (torch.randint(0, VOCAB, (2, 10)),), f, dynamo=False, opset_version=17, input_names=["input_ids"], output_names=["logits"],
                          dynamic_axes={"input_ids": {0: "batch", 1: "seq"}, "logits": {0: "batch", 1: "seq"}})
    data = f.getvalue()
    model = P.parse_model(data)
    assert P.serialize(model) == data
    gen = torch.Generator().manual_seed(1)
    return module, model, torch.randint(0, VOCAB, (48, 20), generator=gen).numpy(), torch.randint(0, VOCAB, (16, 20), generator=gen)


def test_the_export_runs_to_what_the_module_computes(llama):
    module, model, _calib, _eval = llama
    for shape in ((2, 10), (3, 17)):                                           # another batch and length than the export's example
        ids = torch.randint(0, VOCAB, shape)
        with torch.no_grad():
            want = module(ids)
        torch.testing.assert_close(GraphRunner(model, device="cpu")(ids)["logits"], want, rtol=1e-4, atol=1e-5)
    out = q_oracle(model, QConfig(weights=QWeightArgs(dtype=QuantType.QUInt4, group_size=128), ignore=["lm_head"]))
    calls = [n for n in out.graph.node if n.domain]
    # K = 64 or 128 < = 128: the group resolves to the input channels (base.py:72), a power of two >= 16 -> MatMulNBits everywhere
    assert len(calls) == 7 * LAYERS and {n.op_type for n in calls} == {"MatMulNBits"}
    assert [n.op_type for n in out.graph.node if "lm_head" in (n.name or "")] == ["MatMul"]
    assert sum(n.op_type == "MatMul" for n in out.graph.node) == 2 * LAYERS + 1       # the attention products and lm_head


@pytest.mark.gpu
def test_baseline_configurations_on_the_llama_export(llama):
    _module, model, calib, evaluation = llama
    want = GraphRunner(model, device="cuda")(evaluation)["logits"]

    def error(m):
        got = GraphRunner(P.parse_model(P.serialize(m)), device="cuda")(evaluation)["logits"]
        return ((got - want).norm() / want.norm()).item()

    # configuration 2: uint4 groups of 128, RTN -> MatMulNBits; the file is the oracle-provider file
    qc = lambda: QConfig(weights=QWeightArgs(dtype=QuantType.QUInt4, group_size=128), ignore=["lm_head"])      # noqa: E731
    rtn = quantize_model(model, qc())
    assert P.serialize(rtn) == P.serialize(q_oracle(model, qc()))
    # configuration 4: GPTQ int4 groups of 128 (-> 64 here), calibration through the graph on the GPU; q / k / v and gate / up
    # share their Hessians
    make = lambda mode: QConfig(weights=QWeightArgs(dtype=QuantType.QInt4, group_size=128, algorithm=GPTQConfig(block_size=32, mode=mode)),   # noqa: E731
                                calibration_data=calib, calibration_params={"num_samples": 48, "batch_size": 12}, ignore=["lm_head"])
    parity, corrected = quantize_model(model, make("parity")), quantize_model(model, make("corrected"))
    plain = quantize_model(model, QConfig(weights=QWeightArgs(dtype=QuantType.QInt4, group_size=128), ignore=["lm_head"]))
    for m in (parity, corrected):
        calls = [n for n in m.graph.node if n.domain]
        assert len(calls) == 7 * LAYERS and {n.op_type for n in calls} == {"QMatMulWeightsOnlyGrouped"}
    ref = q_oracle(model, make("parity"), runner_device="cuda")
    a = {t.name: P.tensor_to_numpy(t) for t in parity.graph.initializer}
    b = {t.name: P.tensor_to_numpy(t) for t in ref.graph.initializer}
    assert set(a) == set(b)
    for name in a:
        if a[name].dtype == np.int8 and a[name].ndim == 2:
            assert (a[name] != b[name]).mean() < 0.01, name
    e_plain, e_parity, e_corrected = error(plain), error(parity), error(corrected)
    print(f"int4 g64 logits error: RTN {e_plain:.4f}, GPTQ as written {e_parity:.4f}, GPTQ corrected {e_corrected:.4f}")
    assert abs(e_parity - e_plain) < 0.02            # the reference's loop is RTN in effect
    assert e_corrected < e_plain                     # the intended update helps on the calibrated distribution
