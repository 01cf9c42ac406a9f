"""Random configurations through the file path on the GPU: whatever the draw, the file the device path emits is the file the writer
emits with the oracle as numeric provider on the same activations -- byte for byte.  `OQ_TEST_FUZZ_EXAMPLES` scales the number of
draws (default 60), `OQ_TEST_FUZZ_SEED` the seed."""
import os
import random

import numpy as np
import pytest
import torch

from onnx_model_helpers import fixture, q_oracle
from onnx_quantize_amd import QActivationArgs, QConfig, QuantType, QWeightArgs
from onnx_quantize_amd import onnx_proto as P
from onnx_quantize_amd.model_quantize import quantize_model

pytestmark = pytest.mark.gpu

INPUTS = {"mlp_gemm": (64,), "mlp_matmul": (5, 64), "block": (6, 64), "wide_matmul": (3, 256)}


def draw(rng: random.Random):
    name = rng.choice(sorted(INPUTS))
    dtype = rng.choice(["int8", "uint8", "int4", "uint4"])
    strategy = rng.choice(["tensor", "channel", "group"])
    weights = dict(dtype=QuantType.from_string(dtype), strategy=strategy, symmetric=rng.random() < 0.5, reduce_range=rng.random() < 0.25,
                   clip_ratio=rng.choice([1.0, 1.0, 0.95, 0.8]), mse=rng.random() < 0.15)
    if strategy == "group":
        weights["group_size"] = rng.choice([8, 16, 32, 64, 128])
    kw = {}
    acts = "none"
    if dtype in ("int8", "uint8") and strategy != "group":
        acts = rng.choice(["none", "none", "in", "out", "both", "dyn_in", "dyn_both", "qlinear"])
    if acts in ("in", "both", "qlinear"):
        kw["input_activations"] = QActivationArgs(dtype=QuantType.from_string(rng.choice(["int8", "uint8"])), is_static=True,
                                                  symmetric=rng.random() < 0.5, reduce_range=rng.random() < 0.2)
    if acts in ("out", "both", "qlinear"):
        kw["output_activations"] = QActivationArgs(dtype=QuantType.from_string(rng.choice(["int8", "uint8"])), is_static=True,
                                                   symmetric=rng.random() < 0.5)
    if acts in ("dyn_in", "dyn_both"):
        kw["input_activations"] = QActivationArgs(dtype=QuantType.QUInt8, is_static=False)
    if acts == "dyn_both":
        kw["output_activations"] = QActivationArgs(dtype=QuantType.QUInt8, is_static=False)
    if acts == "qlinear":
        kw["format"] = "qlinear"
    if acts in ("in", "out", "both", "qlinear"):
        samples = rng.choice([7, 16, 30])
        kw["calibration_params"] = {"num_samples": samples, "batch_size": rng.choice([1, 4, 10, 64]), "momentum": rng.choice([0.0, 0.0, 0.3])}
        kw["calibration_data"] = np.random.default_rng(rng.randrange(1 << 30)).standard_normal((samples, *INPUTS[name])).astype(np.float32) * rng.choice([0.3, 1.0, 5.0])
    if rng.random() < 0.2:
        kw["ignore"] = [rng.choice(["MatMul$", "/0/", "q|k", "Gemm"])]
    return name, weights, kw


def test_random_configurations_device_file_equals_oracle_file():
    examples = int(os.environ.get("OQ_TEST_FUZZ_EXAMPLES", "60"))
    rng = random.Random(int(os.environ.get("OQ_TEST_FUZZ_SEED", "2025")))
    done = skipped = tied = 0
    for _ in range(examples):
        name, weights, kw = draw(rng)
        try:
            make = lambda: QConfig(weights=QWeightArgs(**weights), **kw)      # noqa: E731
            make()
        except (ValueError, NotImplementedError):                              # a combination the configuration classes refuse
            skipped += 1
            continue
        src = fixture(name)
        got_model, want_model = quantize_model(src, make()), q_oracle(src, make(), runner_device="cuda")
        got, want = P.serialize(got_model), P.serialize(want_model)
        where = (name, weights, {k: v for k, v in kw.items() if k != "calibration_data"})
        if weights["mse"] and got != want:
            # the MSE range search is tolerance-aware by design: a row whose two best candidates tie to 1e-4 may take
            # either; then a few integers / one scale of that row differ.  Everything else must still be the same file.
            a = {t.name: P.tensor_to_numpy(t) for t in got_model.graph.initializer}
            b = {t.name: P.tensor_to_numpy(t) for t in want_model.graph.initializer}
            assert list(a) == list(b), where
            for key in a:
                assert a[key].shape == b[key].shape and a[key].dtype == b[key].dtype, (where, key)
                assert (a[key] != b[key]).mean() < 0.02, (where, key)
            tied += 1
        else:
            assert got == want, where
        done += 1
    print(f"{done - tied} configurations compared as bytes, {tied} with mse = True within the search's tie band, {skipped} refused by the configuration classes")
    assert done >= examples // 2, (done, skipped)
    torch.cuda.synchronize()
