"""The reference's integration tests (test/integration/bert/: DistilBERT for sequence classification, exported to ONNX, quantized
under twelve configurations, evaluated) on a DistilBERT of the same architecture at toy width.

No checkpoint or dataset is reachable offline, so the model has random weights (`transformers.DistilBertForSequenceClassification`
from a config) and is exported here by torch's own ONNX exporter -- a real architecture export with its Where / Expand / Equal mask
plumbing, Erf-Gelu, LayerNormalization, Gemm classifier head -- and "accuracy" becomes agreement with the float model: the share of
inputs whose predicted class is unchanged, next to the relative error of the logits.  Thresholds are the measured values with
margin; for the configurations whose arithmetic is exact the emitted file is also compared, byte for byte, with the file the
writer emits with the oracle as numeric provider on the same activations.
"""
import io
import warnings

import pytest
import torch

from onnx_model_helpers import q_oracle
from onnx_quantize_amd import AwqConfig, HqqConfig, QActivationArgs, QConfig, QuantType, QWeightArgs, SmoothQuantConfig
from onnx_quantize_amd import onnx_proto as P
from onnx_quantize_amd.graph_runner import GraphRunner
from onnx_quantize_amd.model_quantize import quantize_model

DIM, HIDDEN, VOCAB, LAYERS = 128, 256, 120, 2


@pytest.fixture(scope="module")
def distilbert():
    transformers = pytest.importorskip("transformers")
    from torch.onnx._internal.torchscript_exporter import onnx_proto_utils
    onnx_proto_utils._add_onnxscript_fn = lambda proto, _ops: proto          # the exporter's only use of the `onnx` package
    torch.manual_seed(0)
    cfg = transformers.DistilBertConfig(vocab_size=VOCAB, dim=DIM, n_layers=LAYERS, n_heads=4, hidden_dim=HIDDEN, max_position_embeddings=48,
                                        num_labels=2, attn_implementation="eager")
    module = transformers.DistilBertForSequenceClassification(cfg).eval()
    with torch.no_grad():                                                     # a head that separates the classes a little
        module.classifier.weight.mul_(8.0)
    ids, mask = torch.randint(0, VOCAB, (3, 12)), torch.ones(3, 12, dtype=torch.int64)
    f = io.BytesIO()
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        torch.onnx.export(module, (ids, mask), f, dynamo=False, opset_version=17, input_names=["input_ids", "attention_mask"], output_names=["logits"],
                          dynamic_axes={"input_ids": {0: "batch", 1: "seq"}, "attention_mask": {0: "batch", 1: "seq"}, "logits": {0: "batch"}})
    data = f.getvalue()
    model = P.parse_model(data)
    assert P.serialize(model) == data                                          # another producer's file, byte for byte
    gen = torch.Generator().manual_seed(1)

    def samples(n, length):
        ids = torch.randint(0, VOCAB, (n, length), generator=gen)
        mask = torch.ones(n, length, dtype=torch.int64)
        for i in range(n):                                                     # right padding like a tokenizer's batch
            mask[i, int(torch.randint(length // 2, length + 1, (1,), generator=gen)):] = 0
        return {"input_ids": ids.numpy(), "attention_mask": mask.numpy()}

    return module, model, samples(32, 24), samples(100, 24)


def test_the_export_runs_to_what_the_module_computes(distilbert):
    module, model, _calib, evaluation = distilbert
    feed = {k: torch.from_numpy(v) for k, v in evaluation.items()}
    with torch.no_grad():
        want = module(feed["input_ids"], feed["attention_mask"]).logits
    got = GraphRunner(model, device="cpu")(feed)["logits"]
    torch.testing.assert_close(got, want, rtol=1e-4, atol=1e-5)
    # the 14 weights with constant operands are targets; the two attention products of each layer are not
    out = q_oracle(model, QConfig(weights=QWeightArgs(dtype=QuantType.QUInt8, strategy="channel")))
    quantized = [n for n in out.graph.node if n.domain == "quant"]
    assert len(quantized) == 6 * LAYERS + 2 and sum(n.op_type == "MatMul" for n in out.graph.node) == 2 * LAYERS
    assert {n.op_type for n in quantized} == {"QGemmWeightsOnlyQDQ", "QMatMulWeightsOnlyQDQ"}
    # (the pre-classifier's zero bias is an Identity of another layer's in this export: a target only once that is eliminated)


def _w(dtype, **kw):
    return QWeightArgs(dtype=dtype, **kw)


def _a(dtype, static):
    return QActivationArgs(dtype=dtype, is_static=static)


U8, I8, U4 = QuantType.QUInt8, QuantType.QInt8, QuantType.QUInt4
# (id, weights, input activations, output activations, preprocessor, exact file?, min agreement, max logits error)
CASES = [
    # test_bert_weights_only.py:11-17
    ("w_uint8_channel", lambda: _w(U8, symmetric=False, strategy="channel"), None, None, None, True, 0.97, 0.02),
    ("w_uint4_g128", lambda: _w(U4, symmetric=False, strategy="group", group_size=128), None, None, None, True, 0.85, 0.25),
    ("w_uint4_g128_hqq", lambda: _w(U4, symmetric=False, strategy="group", group_size=128, algorithm=HqqConfig(early_stop=False)), None, None, None, False, 0.85, 0.25),
    ("w_int8_channel_awq", lambda: _w(I8, symmetric=False, strategy="channel"), None, None, lambda: AwqConfig(), False, 0.95, 0.05),
    # test_bert_weights_inputs.py:12-18
    ("wi_dynamic", lambda: _w(U8, symmetric=False, strategy="channel"), lambda: _a(U8, False), None, None, True, 0.95, 0.03),
    ("wi_static_smooth", lambda: _w(U8, symmetric=False, strategy="channel"), lambda: _a(U8, True), None, lambda: SmoothQuantConfig(alpha=0.5), False, 0.95, 0.1),
    ("wi_static_awq_clip", lambda: _w(U8, symmetric=False, strategy="channel"), lambda: _a(U8, True), None, lambda: AwqConfig(clip_search=True), False, 0.95, 0.06),
    ("wi_static_int8_sym", lambda: _w(I8, symmetric=True, strategy="channel"), lambda: _a(I8, True), None, None, True, 0.95, 0.035),
    # test_bert_weights_inputs_outputs.py:19-25
    ("wio_dynamic", lambda: _w(U8, symmetric=False, strategy="channel"), lambda: _a(U8, False), lambda: _a(U8, False), None, True, 0.95, 0.035),
    ("wio_static_smooth", lambda: _w(U8, symmetric=False, strategy="channel"), lambda: _a(U8, True), lambda: _a(U8, True), lambda: SmoothQuantConfig(alpha=0.5), False, 0.95, 0.1),
    ("wio_static_awq", lambda: _w(U8, symmetric=False, strategy="channel"), lambda: _a(U8, True), lambda: _a(U8, True), lambda: AwqConfig(), False, 0.95, 0.06),
    ("wio_static_int8_sym", lambda: _w(I8, symmetric=True, strategy="channel"), lambda: _a(I8, True), lambda: _a(I8, True), None, True, 0.95, 0.045),
]


def _config(case, calib):
    _id, w, i, o, pre, _exact, _agree, _err = case
    return QConfig(weights=w(), input_activations=i() if i else None, output_activations=o() if o else None,
                   preprocessors=[pre()] if pre else [], calibration_data=calib)


@pytest.mark.gpu
@pytest.mark.parametrize("case", CASES, ids=[c[0] for c in CASES])
def test_quantize_distilbert(distilbert, case):
    _module, model, calib, evaluation = distilbert
    out = quantize_model(model, _config(case, calib))
    data = P.serialize(out)
    quantized = [n for n in out.graph.node if n.domain in ("quant", "com.microsoft")]
    assert len(quantized) == 6 * LAYERS + 2
    if case[5]:
        assert data == P.serialize(q_oracle(model, _config(case, calib), runner_device="cuda")), case[0]
    feed = {k: torch.from_numpy(v) for k, v in evaluation.items()}
    want = GraphRunner(model, device="cuda")(feed)["logits"]
    got = GraphRunner(P.parse_model(data), device="cuda")(feed)["logits"]
    agreement = (got.argmax(-1) == want.argmax(-1)).float().mean().item()
    error = ((got - want).norm() / want.norm()).item()
    print(f"{case[0]}: agreement {agreement:.2f}, logits rel err {error:.4f}")
    assert agreement >= case[6] and error <= case[7], (case[0], agreement, error)
