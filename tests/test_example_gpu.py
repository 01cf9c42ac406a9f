"""The device-resident example (examples/gemma3_shapes/) stays runnable: two blocks, three batches, results of the batched
driver path against the per-layer calls."""
import importlib.util
import os

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def test_gemma3_shaped_example_runs_and_matches_the_per_layer_path():
    import torch

    from onnx_quantize_amd.hip import ops

    spec = importlib.util.spec_from_file_location("gemma3_shapes_gptq", os.path.join(ROOT, "examples", "gemma3_shapes", "gemma3_shapes_gptq.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    out, blobs, quantized, weights = mod.main(layers=2, batches=3, verbose=False)
    assert out["weights"] == 8 and out["activation_qparams"] == 16 and out["blob_shape_of_0.qkv"] == (1024, 5, 64)
    # parity mode: the integers are those of the fused RTN kernel on the same weight, the blob is its packing
    for name in ("0.qkv", "1.down"):
        w = weights[name][0]
        rq, rs, rz = ops.rtn_quantize(w, "uint4", "group", 128)
        q, s, z, info = quantized[name]
        assert int(info.item()) == 0
        assert torch.equal(q, rq) and torch.equal(z.reshape(-1), rz.reshape(-1))
        blob, _, _ = ops.rtn_quantize(w, "uint4", "group", 128, layout="nbits")
        assert torch.equal(blobs[name][0], blob)


def test_gemma3_onnx_file_example_runs_the_references_two_configurations(tmp_path):
    """examples/gemma3_shapes/gemma3_onnx_file.py at toy size: a genai-builder-style file (GroupQueryAttention with rotary
    caches and empty past_key_values, Skip / SimplifiedLayerNormalization) through the reference's gemma3_rtn.py and
    gemma3_awq.py configurations, file to file."""
    from onnx_quantize_amd import onnx_proto as P

    spec = importlib.util.spec_from_file_location("gemma3_onnx_file", os.path.join(ROOT, "examples", "gemma3_shapes", "gemma3_onnx_file.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    report = mod.main(layers=2, vocab=512, samples=4, block=32, work=str(tmp_path), verbose=False, static_samples=8)
    rtn, awq = report["rtn_int8_g128"], report["awq_uint4_g128"]
    assert rtn["calls"] == {"QMatMulWeightsOnlyGrouped": 14} and rtn["lm_head_left_float"] and rtn["logits_rel_err"] < 0.05
    assert awq["calls"] == {"MatMulNBits": 14} and awq["lm_head_left_float"] and awq["logits_rel_err"] < 0.35
    static = report["static_int8_8"]
    assert static["calls"] == {"QMatMulWeightStaticInputOutputQDQ": 14} and static["logits_rel_err"] < 0.2
    q = P.load_model(tmp_path / "qgemma_awq_uint4_g128.onnx")
    muls = [n for n in q.graph.node if n.op_type == "Mul" and n.name.endswith("/scale_input")]
    assert len(muls) == 14                                     # awq.py:73-88: one Mul per quantized node
    assert all(t.data_location is None for t in q.graph.initializer) and (tmp_path / "qgemma_awq_uint4_g128.onnx.data").exists()
    # K = 640 / 1024 / 2048 with g = 128: power-of-two blocks -> MatMulNBits with nibble-packed zero points
    node = next(n for n in q.graph.node if n.name == "/model/layers.1/mlp/down_proj/MatMul")
    assert {a.name: P.attribute_value(a) for a in node.attribute} == dict(K=2048, N=640, bits=4, block_size=128)
