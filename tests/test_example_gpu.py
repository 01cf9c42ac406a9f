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
