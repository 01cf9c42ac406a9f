"""A ViT-architecture export through the writer: a strided Conv as patch embedding, a class token concatenated in front, learned
position embeddings, pre-norm transformer blocks with biased projections (MatMul + Add on 3-D values: they stay MatMuls, the
rank-2 rule of `matmul_add_to_gemm` does not fire), and a classifier on the class token (a rank-2 Linear: a Gemm).  One image
input; the calibration walk has to get through the convolution and the token plumbing to reach the projections.

A plain torch restatement, exported here by torch's ONNX exporter (another producer's file)."""
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

DIM, HEADS, LAYERS, PATCH, SIDE, CLASSES = 64, 4, 2, 4, 16, 10


class Block(torch.nn.Module):
    def __init__(self):
        super().__init__()
        self.ln1, self.ln2 = torch.nn.LayerNorm(DIM), torch.nn.LayerNorm(DIM)
        self.qkv, self.proj = torch.nn.Linear(DIM, 3 * DIM), torch.nn.Linear(DIM, DIM)
        self.fc1, self.fc2 = torch.nn.Linear(DIM, 2 * DIM), torch.nn.Linear(2 * DIM, DIM)

    def forward(self, x):
        b, t, _ = x.shape
        q, k, v = self.qkv(self.ln1(x)).reshape(b, t, 3, HEADS, DIM // HEADS).permute(2, 0, 3, 1, 4)
        a = torch.softmax(q @ k.transpose(-2, -1) / math.sqrt(DIM // HEADS), dim=-1)
        x = x + self.proj((a @ v).transpose(1, 2).reshape(b, t, DIM))
        return x + self.fc2(torch.nn.functional.gelu(self.fc1(self.ln2(x))))


class ViT(torch.nn.Module):
    def __init__(self):
        super().__init__()
        self.patch = torch.nn.Conv2d(3, DIM, PATCH, stride=PATCH)
        self.cls = torch.nn.Parameter(torch.randn(1, 1, DIM) * 0.02)
        self.pos = torch.nn.Parameter(torch.randn(1, (SIDE // PATCH) ** 2 + 1, DIM) * 0.02)
        self.blocks = torch.nn.ModuleList([Block() for _ in range(LAYERS)])
        self.norm, self.head = torch.nn.LayerNorm(DIM), torch.nn.Linear(DIM, CLASSES)

    def forward(self, image):
        x = self.patch(image).flatten(2).transpose(1, 2)
        x = torch.cat((self.cls.expand(x.shape[0], -1, -1), x), dim=1) + self.pos
        for block in self.blocks:
            x = block(x)
        return self.head(self.norm(x)[:, 0])


@pytest.fixture(scope="module")
def vit():
    from torch.onnx._internal.torchscript_exporter import onnx_proto_utils
    onnx_proto_utils._add_onnxscript_fn = lambda proto, _ops: proto
    torch.manual_seed(0)
    module = ViT().eval()
    f = io.BytesIO()
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        torch.onnx.export(module, (torch.randn(2, 3, SIDE, SIDE),), f, dynamo=False, opset_version=17, input_names=["image"], output_names=["scores"],
                          dynamic_axes={"image": {0: "batch"}, "scores": {0: "batch"}})
    data = f.getvalue()
    model = P.parse_model(data)
    assert P.serialize(model) == data
    gen = torch.Generator().manual_seed(1)
    return module, model, torch.randn(24, 3, SIDE, SIDE, generator=gen).numpy(), torch.randn(8, 3, SIDE, SIDE, generator=gen)


def test_the_export_runs_and_its_projections_and_head_are_targets(vit):
    module, model, _calib, evaluation = vit
    with torch.no_grad():
        want = module(evaluation)
    torch.testing.assert_close(GraphRunner(model, device="cpu")(evaluation)["scores"], want, rtol=1e-4, atol=1e-5)
    assert {"Conv", "Concat", "LayerNormalization", "Erf", "Gemm"} <= {n.op_type for n in model.graph.node}
    targets = target_nodes(model, QConfig(weights=QWeightArgs()))
    assert sorted(t[1] for t in targets) == ["Gemm"] + ["MatMul"] * (4 * LAYERS)        # 3-D projections stay MatMuls; the head is a Gemm
    out = q_oracle(model, QConfig(weights=QWeightArgs(dtype=QuantType.QInt8, strategy="channel")))
    assert sum(n.op_type == "Conv" for n in out.graph.node) == 1 and sum(bool(n.domain) for n in out.graph.node) == 4 * LAYERS + 1
    got = GraphRunner(out, device="cpu")(evaluation)["scores"]
    assert ((got - want).norm() / want.norm()).item() < 0.05


@pytest.mark.gpu
def test_device_files_equal_the_oracle_files_on_the_vit_export(vit):
    _module, model, calib, evaluation = vit
    want = GraphRunner(model, device="cuda")(evaluation)["scores"]
    act = lambda: QActivationArgs(dtype=QuantType.QUInt8, is_static=True)      # noqa: E731
    configs = {
        "uint4_g32": lambda: QConfig(weights=QWeightArgs(dtype=QuantType.QUInt4, group_size=32)),
        "static_in_out": lambda: QConfig(weights=QWeightArgs(dtype=QuantType.QInt8, strategy="channel"), input_activations=act(), output_activations=act(),
                                         calibration_data=calib, calibration_params={"num_samples": 24, "batch_size": 8}),
        "static_qlinear": lambda: QConfig(weights=QWeightArgs(dtype=QuantType.QInt8, symmetric=True), format="qlinear", input_activations=act(),
                                          output_activations=act(), calibration_data=calib, calibration_params={"num_samples": 24, "batch_size": 8}),
    }
    for name, make in configs.items():
        data = P.serialize(quantize_model(model, make()))
        assert data == P.serialize(q_oracle(model, make(), runner_device="cuda")), name
        got = GraphRunner(P.parse_model(data), device="cuda")(evaluation)["scores"]
        error = ((got - want).norm() / want.norm()).item()
        print(f"{name}: scores rel err {error:.4f}")
        assert error < 0.3, (name, error)
