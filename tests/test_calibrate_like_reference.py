"""The reference's tests of `calibrate_model` (test/core/calibration/test_calibrate.py) on the on-device calibration walk of the
standalone writer (`model_quantize._calibrate`: `GraphRunner` on torch-ROCm -> `ActivationStream` -> the HIP reductions).
Same toy models, same configurations, same assertions on what ends up in the nodes' metadata; where the reference only checks
that a key exists, the value is also compared with the oracle's calibrator on the activations of the same run."""
import numpy as np
import pytest
import torch

import oq_oracle as O
from onnx_quantize_amd import CalibrationParams, GPTQConfig, QActivationArgs, QConfig, QuantType, QWeightArgs
from onnx_quantize_amd import onnx_proto as P
from onnx_quantize_amd.graph_runner import GraphRunner
from onnx_quantize_amd.model_quantize import _calibrate, _Graph, _target_nodes
from onnx_quantize_amd.reference_passes import StreamedGptqInput

pytestmark = pytest.mark.gpu


def _truncated_normal(rng, shape, scale=0.1, clip=2.5):
    return np.clip(rng.normal(0.0, scale, size=shape), -clip * scale, clip * scale).astype(np.float32)


def _model(nodes, inits, inputs, outputs=("Y",), opset=20):
    g = P.Message("GraphProto", name="test_model", node=nodes, initializer=[P.numpy_to_tensor(k, v) for k, v in inits.items()],
                  input=[P.make_value_info(n, t, shape) for n, t, shape in inputs], output=[P.make_value_info(o, 1, None) for o in outputs])
    return P.Message("ModelProto", ir_version=10, graph=g, opset_import=[P.Message("OperatorSetIdProto", domain="", version=opset)])


def single_input_model(rng):                                                  # test_calibrate.py:35-49
    return _model([P.make_node("MatMul", ["X", "W1"], ["x1"], name="fc1"), P.make_node("Relu", ["x1"], ["x2"], name="relu"),
                   P.make_node("MatMul", ["x2", "W2"], ["Y"], name="fc2")],
                  {"W1": _truncated_normal(rng, (32, 64)), "W2": _truncated_normal(rng, (64, 128))}, [("X", 1, ["N", 32])])


def multi_input_model(rng):                                                   # :16-32 (W1 is read by two MatMuls)
    return _model([P.make_node("MatMul", ["X", "W1"], ["x1"], name="fx"), P.make_node("MatMul", ["Z", "W1"], ["z1"], name="fz"),
                   P.make_node("Add", ["x1", "z1"], ["x2"], name="add"), P.make_node("Relu", ["x2"], ["x3"], name="relu"),
                   P.make_node("MatMul", ["x3", "W2"], ["Y"], name="fc")],
                  {"W1": _truncated_normal(rng, (32, 64)), "W2": _truncated_normal(rng, (64, 128))}, [("X", 1, ["N", 32]), ("Z", 1, ["N", 32])])


def _calibrated(model, qconfig):
    G = _Graph(model.graph)
    targets = _target_nodes(G, qconfig)
    return targets, _calibrate(model, G, targets, qconfig, "cuda")


def _expect_keys(model, qconfig, keys):
    targets, meta = _calibrated(model, qconfig)
    assert len(targets) == sum(n.op_type in qconfig.target_op_types for n in model.graph.node)
    for n in targets:
        assert set(keys) <= set(meta[id(n)]), (n.name, meta[id(n)].keys())
    return targets, meta


def _act():
    return QActivationArgs(dtype=QuantType.QUInt8, is_static=True)


@pytest.mark.parametrize("num_samples", [1, 10])
@pytest.mark.parametrize("kinds", [("input",), ("output",), ("input", "output")])
def test_calibrate_model_with_samples(num_samples, kinds):                    # :52-106
    rng = np.random.default_rng(num_samples)
    data = _truncated_normal(rng, (num_samples, 32))
    qc = QConfig(weights=QWeightArgs(dtype=QuantType.QUInt8), calibration_data=data, **{f"{k}_activations": _act() for k in kinds})
    model = single_input_model(rng)
    targets, meta = _expect_keys(model, qc, [f"{k}_{p}" for k in kinds for p in ("scale", "zero_point")])
    # the values: the oracle's calibrator on the values the same graph produces for the same batches
    names = [n.input[0] for n in targets] + [n.output[0] for n in targets]
    runner = GraphRunner(model, outputs=list(dict.fromkeys(names)), device="cuda")
    acts = [{k: v.cpu().numpy() for k, v in runner(torch.from_numpy(np.ascontiguousarray(b))).items()} for b in O.prepare_calibration_data(data, 10, 100)]
    key = ("uint8", False, False)
    ins, outs = [n.input[0] for n in targets], [n.output[0] for n in targets]
    flow = O.calibrate_flow([{k: b[k] for k in (ins if "input" in kinds else []) + (outs if "output" in kinds else [])} for b in acts],
                            ins if "input" in kinds else [], outs if "output" in kinds else [], 0.0,
                            key if "input" in kinds else None, key if "output" in kinds else None)
    for n in targets:
        for kind, name in (("input", n.input[0]), ("output", n.output[0])):
            if kind in kinds:
                s, z = flow[(kind, name)]
                assert meta[id(n)][f"{kind}_scale"].tobytes() == np.asarray(s, np.float32).tobytes()
                assert int(meta[id(n)][f"{kind}_zero_point"]) == int(z) and meta[id(n)][f"{kind}_zero_point"].dtype == np.uint8


def test_calibrate_model_random_samples():                                    # :109-121
    qc = QConfig(weights=QWeightArgs(dtype=QuantType.QUInt8), input_activations=_act())
    _expect_keys(single_input_model(np.random.default_rng(0)), qc, ["input_scale", "input_zero_point"])


def test_calibrate_model_gptq():                                              # :124-136
    rng = np.random.default_rng(1)
    qc = QConfig(weights=QWeightArgs(dtype=QuantType.QUInt8, algorithm=GPTQConfig()), calibration_data=_truncated_normal(rng, (10, 32)))
    targets, meta = _expect_keys(single_input_model(rng), qc, ["input"])
    for n in targets:                                                         # the Hessian of the node's input instead of the input itself
        got = meta[id(n)]["input"]
        assert isinstance(got, StreamedGptqInput) and got.n == 10 and got.h.shape[0] == {"fc1": 32, "fc2": 64}[n.name]


@pytest.mark.parametrize("batch_size,num_samples", [(2, 10), (5, 10), (10, 10), (20, 10), (3, 10)])
def test_calibrate_model_with_batch_size(batch_size, num_samples):            # :139-176 (3: one sample dropped)
    rng = np.random.default_rng(batch_size)
    qc = QConfig(weights=QWeightArgs(dtype=QuantType.QUInt8), input_activations=_act(), calibration_data=_truncated_normal(rng, (num_samples, 32)),
                 calibration_params=CalibrationParams(batch_size=batch_size, num_samples=num_samples))
    _expect_keys(single_input_model(rng), qc, ["input_scale", "input_zero_point"])


def test_calibrate_model_random_with_batch_size():                            # :179-193
    qc = QConfig(weights=QWeightArgs(dtype=QuantType.QUInt8), input_activations=_act(), calibration_params={"batch_size": 4, "num_samples": 12})
    _expect_keys(single_input_model(np.random.default_rng(2)), qc, ["input_scale", "input_zero_point"])


def test_calibrate_model_multi_input():                                       # :196-238
    rng = np.random.default_rng(3)
    qc = QConfig(weights=QWeightArgs(dtype=QuantType.QUInt8), input_activations=_act())
    _expect_keys(multi_input_model(rng), qc, ["input_scale", "input_zero_point"])
    data = {"X": _truncated_normal(rng, (10, 32)), "Z": _truncated_normal(rng, (10, 32))}
    qc = QConfig(weights=QWeightArgs(dtype=QuantType.QUInt8), input_activations=_act(), calibration_data=data)
    _expect_keys(multi_input_model(rng), qc, ["input_scale", "input_zero_point"])
    qc = QConfig(weights=QWeightArgs(dtype=QuantType.QUInt8), input_activations=_act(), calibration_data=_truncated_normal(rng, (10, 32)))
    with pytest.raises(ValueError, match="Calibration data must be a dict"):
        _calibrated(multi_input_model(rng), qc)


def test_calibrate_model_random_samples_int32_input():                        # :241-265: token ids -> Gather -> MatMul
    rng = np.random.default_rng(4)
    model = _model([P.make_node("Gather", ["W_embed", "input_ids"], ["x_emb"], name="emb"), P.make_node("MatMul", ["x_emb", "W1"], ["Y"], name="fc")],
                   {"W_embed": _truncated_normal(rng, (100, 64)), "W1": _truncated_normal(rng, (64, 128))}, [("input_ids", P.DataType.INT32, ["N", "S"])])
    qc = QConfig(weights=QWeightArgs(dtype=QuantType.QUInt8), input_activations=_act())
    _expect_keys(model, qc, ["input_scale", "input_zero_point"])
