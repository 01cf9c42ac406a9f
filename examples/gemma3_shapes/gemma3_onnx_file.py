#!/usr/bin/env python3
"""The reference's gemma3 examples (examples/gemma3/gemma3_rtn.py, gemma3_awq.py) on an ONNX FILE, without the ONNX stack.

The reference exports gemma-3-270m with onnxruntime-genai's model builder, loads the file with `onnx_ir`, calls
`quantize(model, qconfig)` and saves with external data.  Offline there is neither the checkpoint nor that stack, so this
script writes a file of the same architecture and style itself -- gemma-3-270m's dimensions (hidden 640, 18 layers, 4 query
heads / 1 key-value head of 256, MLP 2048, q / k norms, four RMS norms per layer, 5 local layers per global one, tied
vocabulary of 262144) in the builder's vocabulary: `com.microsoft::GroupQueryAttention` with rotary caches and
`past_key_values.*` inputs, `SimplifiedLayerNormalization` / `SkipSimplifiedLayerNormalization`, `/model/layers.N/attn/q_proj/MatMul`
node names, `/lm_head/MatMul` -- with random weights, and then runs the reference's two example configurations on it:

  gemma3_rtn.py   QConfig(weights=QWeightArgs(dtype="int8", strategy="group", group_size=128), ignore=["lm_head"])
  gemma3_awq.py   QConfig(weights=QWeightArgs(dtype="uint4", strategy="group", group_size=128), preprocessors=[AwqConfig()],
                          calibration_data={input_ids, attention_mask, empty past_key_values}, CalibrationParams(batch_size=1,
                          num_samples=64), ignore=["lm_head"])
  BASELINE #3     static int8 weights and activations (QDQ), 512 calibration sequences (the configuration this repository's
                  BASELINE.json names for the gemma3 example)

Everything numeric happens on the GPU: the AWQ calibration runs the graph itself on torch-ROCm (`GraphRunner`), the 126
searches and the quantization run in the HIP library, and the result is written with its tensors in a side file.

    python examples/gemma3_shapes/gemma3_onnx_file.py                       # full size, both configurations
    python examples/gemma3_shapes/gemma3_onnx_file.py --layers 2 --vocab 512 --samples 4 --block 32
"""
import argparse
import json
import os
import shutil
import sys
import tempfile
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from onnx_quantize_amd import AwqConfig, CalibrationParams, QConfig, QWeightArgs  # noqa: E402
from onnx_quantize_amd import onnx_proto as P  # noqa: E402
from onnx_quantize_amd.graph_runner import GraphRunner  # noqa: E402
from onnx_quantize_amd.model_quantize import quantize_model  # noqa: E402

HIDDEN, HEADS, KV_HEADS, HEAD_DIM, FFN, WINDOW = 640, 4, 1, 256, 2048, 512


def build_model(layers=18, vocab=262144, max_positions=2048, seed=0) -> P.Message:
    gen = torch.Generator().manual_seed(seed)
    nodes, inits = [], []
    ms = "com.microsoft"

    def tensor(name, value):
        inits.append(P.numpy_to_tensor(name, value))
        return name

    def weight(name, k, n):
        return tensor(name, (torch.randn(k, n, generator=gen) * (1.0 / k ** 0.5)).numpy())

    def norm_weight(name, width):
        return tensor(name, (1.0 + 0.1 * torch.randn(width, generator=gen)).numpy())     # the builder folds gemma's "+ 1" in

    for tag, theta in (("local", 10000.0), ("global", 1000000.0)):
        inv = 1.0 / (theta ** (torch.arange(0, HEAD_DIM, 2).float() / HEAD_DIM))
        freqs = torch.outer(torch.arange(max_positions).float(), inv)
        tensor(f"cos_cache_{tag}", freqs.cos().numpy())
        tensor(f"sin_cache_{tag}", freqs.sin().numpy())
    tensor("/model/constants/one", np.array([1], dtype=np.int64))
    tensor("/model/constants/axis1", np.array([1], dtype=np.int64))
    tensor("/model/constants/idx1", np.array(1, dtype=np.int64))
    tensor("/model/constants/embed_scale", np.array(HIDDEN ** 0.5, dtype=np.float32))
    tensor("/model/constants/q_heads_shape", np.array([0, 0, HEADS, HEAD_DIM], dtype=np.int64))
    tensor("/model/constants/kv_heads_shape", np.array([0, 0, KV_HEADS, HEAD_DIM], dtype=np.int64))
    tensor("/model/constants/q_flat_shape", np.array([0, 0, HEADS * HEAD_DIM], dtype=np.int64))
    tensor("/model/constants/kv_flat_shape", np.array([0, 0, KV_HEADS * HEAD_DIM], dtype=np.int64))
    tensor("model.embed_tokens.weight", (torch.randn(vocab, HIDDEN, generator=gen) * 0.05).numpy())

    # attention-mask bookkeeping of the builder: seqlens_k = sum(mask) - 1, total_sequence_length = mask.shape[1]
    nodes += [
        P.make_node("ReduceSum", ["attention_mask", "/model/constants/axis1"], ["/model/attn_mask_reformat/ReduceSum/out"], name="/model/attn_mask_reformat/ReduceSum", keepdims=0),
        P.make_node("Sub", ["/model/attn_mask_reformat/ReduceSum/out", "/model/constants/one"], ["/model/attn_mask_reformat/Sub/out"], name="/model/attn_mask_reformat/Sub"),
        P.make_node("Cast", ["/model/attn_mask_reformat/Sub/out"], ["seqlens_k"], name="/model/attn_mask_reformat/Cast", to=P.DataType.INT32),
        P.make_node("Shape", ["attention_mask"], ["/model/attn_mask_reformat/Shape/out"], name="/model/attn_mask_reformat/Shape"),
        P.make_node("Gather", ["/model/attn_mask_reformat/Shape/out", "/model/constants/idx1"], ["/model/attn_mask_reformat/Gather/out"], name="/model/attn_mask_reformat/Gather"),
        P.make_node("Cast", ["/model/attn_mask_reformat/Gather/out"], ["total_seq_len"], name="/model/attn_mask_reformat/Cast_1", to=P.DataType.INT32),
        P.make_node("Gather", ["model.embed_tokens.weight", "input_ids"], ["/model/embed_tokens/Gather/out"], name="/model/embed_tokens/Gather"),
        P.make_node("Mul", ["/model/embed_tokens/Gather/out", "/model/constants/embed_scale"], ["/model/embed_tokens/Mul/out"], name="/model/embed_tokens/Mul"),
    ]
    inputs = [P.make_value_info("input_ids", P.DataType.INT64, ["batch_size", "sequence_length"]),
              P.make_value_info("attention_mask", P.DataType.INT64, ["batch_size", "total_sequence_length"])]
    outputs = [P.make_value_info("logits", P.DataType.FLOAT, ["batch_size", "sequence_length", vocab])]
    residual, branch = "/model/embed_tokens/Mul/out", None
    for i in range(layers):
        p = f"/model/layers.{i}"
        w = f"model.layers.{i}"
        normed = f"{p}/input_layernorm/out"
        if branch is None:
            nodes.append(P.make_node("SimplifiedLayerNormalization", [residual, norm_weight(f"{w}.input_layernorm.weight", HIDDEN)], [normed],
                                     name=f"{p}/input_layernorm/LayerNorm", axis=-1, epsilon=1e-6))
        else:
            nodes.append(P.make_node("SkipSimplifiedLayerNormalization", [residual, branch, norm_weight(f"{w}.input_layernorm.weight", HIDDEN)],
                                     [normed, "", "", f"{p}/input_layernorm/sum"], name=f"{p}/input_layernorm/SkipLayerNorm", domain=ms, epsilon=1e-6))
            residual = f"{p}/input_layernorm/sum"
        for proj, width in (("q", HEADS * HEAD_DIM), ("k", KV_HEADS * HEAD_DIM), ("v", KV_HEADS * HEAD_DIM)):
            nodes.append(P.make_node("MatMul", [normed, weight(f"{w}.attn.{proj}_proj.MatMul.weight", HIDDEN, width)], [f"{p}/attn/{proj}_proj/MatMul/out"],
                                     name=f"{p}/attn/{proj}_proj/MatMul"))
        for proj, kind in (("q", "q"), ("k", "kv")):                          # gemma3's per-head q / k norms
            nodes += [
                P.make_node("Reshape", [f"{p}/attn/{proj}_proj/MatMul/out", f"/model/constants/{kind}_heads_shape"], [f"{p}/attn/{proj}_norm/heads"], name=f"{p}/attn/{proj}_norm/Reshape_1"),
                P.make_node("SimplifiedLayerNormalization", [f"{p}/attn/{proj}_norm/heads", norm_weight(f"{w}.attn.{proj}_norm.weight", HEAD_DIM)],
                            [f"{p}/attn/{proj}_norm/normed"], name=f"{p}/attn/{proj}_norm/LayerNorm", axis=-1, epsilon=1e-6),
                P.make_node("Reshape", [f"{p}/attn/{proj}_norm/normed", f"/model/constants/{kind}_flat_shape"], [f"{p}/attn/{proj}_norm/out"], name=f"{p}/attn/{proj}_norm/Reshape_2"),
            ]
        local = (i + 1) % 6 != 0
        tag = "local" if local else "global"
        gqa_attrs = dict(num_heads=HEADS, kv_num_heads=KV_HEADS, do_rotary=1, scale=float(HEAD_DIM ** -0.5))
        if local:
            gqa_attrs["local_window_size"] = WINDOW
        nodes.append(P.make_node("GroupQueryAttention", [f"{p}/attn/q_norm/out", f"{p}/attn/k_norm/out", f"{p}/attn/v_proj/MatMul/out",
                                                          f"past_key_values.{i}.key", f"past_key_values.{i}.value", "seqlens_k", "total_seq_len",
                                                          f"cos_cache_{tag}", f"sin_cache_{tag}"],
                                 [f"{p}/attn/GroupQueryAttention/out", f"present.{i}.key", f"present.{i}.value"], name=f"{p}/attn/GroupQueryAttention", domain=ms, **gqa_attrs))
        nodes.append(P.make_node("MatMul", [f"{p}/attn/GroupQueryAttention/out", weight(f"{w}.attn.o_proj.MatMul.weight", HEADS * HEAD_DIM, HIDDEN)],
                                 [f"{p}/attn/o_proj/MatMul/out"], name=f"{p}/attn/o_proj/MatMul"))
        nodes.append(P.make_node("SimplifiedLayerNormalization", [f"{p}/attn/o_proj/MatMul/out", norm_weight(f"{w}.post_attention_layernorm.weight", HIDDEN)],
                                 [f"{p}/post_attention_layernorm/out"], name=f"{p}/post_attention_layernorm/LayerNorm", axis=-1, epsilon=1e-6))
        nodes.append(P.make_node("SkipSimplifiedLayerNormalization", [residual, f"{p}/post_attention_layernorm/out", norm_weight(f"{w}.pre_feedforward_layernorm.weight", HIDDEN)],
                                 [f"{p}/pre_feedforward_layernorm/out", "", "", f"{p}/pre_feedforward_layernorm/sum"], name=f"{p}/pre_feedforward_layernorm/SkipLayerNorm",
                                 domain=ms, epsilon=1e-6))
        residual = f"{p}/pre_feedforward_layernorm/sum"
        for proj in ("gate", "up"):
            nodes.append(P.make_node("MatMul", [f"{p}/pre_feedforward_layernorm/out", weight(f"{w}.mlp.{proj}_proj.MatMul.weight", HIDDEN, FFN)],
                                     [f"{p}/mlp/{proj}_proj/MatMul/out"], name=f"{p}/mlp/{proj}_proj/MatMul"))
        nodes += [
            P.make_node("Gelu", [f"{p}/mlp/gate_proj/MatMul/out"], [f"{p}/mlp/act_fn/out"], name=f"{p}/mlp/act_fn/Gelu", approximate="tanh"),
            P.make_node("Mul", [f"{p}/mlp/act_fn/out", f"{p}/mlp/up_proj/MatMul/out"], [f"{p}/mlp/Mul/out"], name=f"{p}/mlp/Mul"),
            P.make_node("MatMul", [f"{p}/mlp/Mul/out", weight(f"{w}.mlp.down_proj.MatMul.weight", FFN, HIDDEN)], [f"{p}/mlp/down_proj/MatMul/out"],
                        name=f"{p}/mlp/down_proj/MatMul"),
            P.make_node("SimplifiedLayerNormalization", [f"{p}/mlp/down_proj/MatMul/out", norm_weight(f"{w}.post_feedforward_layernorm.weight", HIDDEN)],
                        [f"{p}/post_feedforward_layernorm/out"], name=f"{p}/post_feedforward_layernorm/LayerNorm", axis=-1, epsilon=1e-6),
        ]
        branch = f"{p}/post_feedforward_layernorm/out"
        inputs += [P.make_value_info(f"past_key_values.{i}.{kv}", P.DataType.FLOAT, ["batch_size", KV_HEADS, "past_sequence_length", HEAD_DIM]) for kv in ("key", "value")]
        outputs += [P.make_value_info(f"present.{i}.{kv}", P.DataType.FLOAT, ["batch_size", KV_HEADS, "total_sequence_length", HEAD_DIM]) for kv in ("key", "value")]
    nodes.append(P.make_node("SkipSimplifiedLayerNormalization", [residual, branch, norm_weight("model.norm.weight", HIDDEN)],
                             ["/model/norm/out"], name="/model/norm/SkipLayerNorm", domain=ms, epsilon=1e-6))
    nodes.append(P.make_node("MatMul", ["/model/norm/out", weight("lm_head.MatMul.weight", HIDDEN, vocab)], ["logits"], name="/lm_head/MatMul"))
    graph = P.Message("GraphProto", name="main_graph", node=nodes, initializer=inits, input=inputs, output=outputs)
    return P.Message("ModelProto", ir_version=10, producer_name="onnx_quantize_amd.examples.gemma3_onnx_file", graph=graph,
                     opset_import=[P.Message("OperatorSetIdProto", domain="", version=21), P.Message("OperatorSetIdProto", domain=ms, version=1)])


def make_calibration_data(layers, vocab, num_samples, block_size, seed=1):
    """gemma3_awq.py:15-39 with random token ids in place of wikitext-2: input_ids, an all-ones mask, EMPTY key / value caches."""
    rng = np.random.default_rng(seed)
    data = {"input_ids": rng.integers(0, vocab, size=(num_samples, block_size), dtype=np.int64),
            "attention_mask": np.ones((num_samples, block_size), dtype=np.int64)}
    empty = np.zeros((num_samples, KV_HEADS, 0, HEAD_DIM), dtype=np.float32)
    for i in range(layers):
        data[f"past_key_values.{i}.key"] = empty
        data[f"past_key_values.{i}.value"] = empty
    return data


def main(layers=18, vocab=262144, samples=64, block=256, work=None, verbose=True, static_samples=512) -> dict:
    keep = work is not None
    work = work or tempfile.mkdtemp(prefix="oq_gemma3_onnx_")
    os.makedirs(work, exist_ok=True)
    src = os.path.join(work, "model.onnx")
    t0 = time.perf_counter()
    P.save_model(build_model(layers, vocab), src, external_data="model.onnx.data")
    t_build = time.perf_counter() - t0
    torch.zeros(1, device="cuda")
    report = {"layers": layers, "vocab": vocab, "build_source_s": round(t_build, 2),
              "source_bytes": os.path.getsize(src) + os.path.getsize(src + ".data")}
    feed = {k: torch.from_numpy(v[:2]) for k, v in make_calibration_data(layers, vocab, 2, min(block, 64), seed=5).items()}
    float_logits = GraphRunner(P.load_model(src), outputs=["logits"], device="cuda")(feed)["logits"]
    configs = {
        "rtn_int8_g128": lambda: QConfig(weights=QWeightArgs(dtype="int8", strategy="group", group_size=128), ignore=["lm_head"]),
        "awq_uint4_g128": lambda: QConfig(weights=QWeightArgs(dtype="uint4", strategy="group", group_size=128), preprocessors=[AwqConfig()],
                                          calibration_data=make_calibration_data(layers, vocab, samples, block),
                                          calibration_params=CalibrationParams(batch_size=1, num_samples=samples), ignore=["lm_head"]),
    }
    if static_samples:
        # BASELINE.json configs[2]: "gemma3 example ONNX, static QInt8 weights + activations with 512-sample calibration"
        from onnx_quantize_amd import QActivationArgs
        configs[f"static_int8_{static_samples}"] = lambda: QConfig(
            weights=QWeightArgs(dtype="int8"), input_activations=QActivationArgs(dtype="int8", is_static=True),
            output_activations=QActivationArgs(dtype="int8", is_static=True), calibration_data=make_calibration_data(layers, vocab, static_samples, block),
            calibration_params=CalibrationParams(batch_size=16, num_samples=static_samples), ignore=["lm_head"])
    for name, make in configs.items():
        dst = os.path.join(work, f"qgemma_{name}.onnx")
        t0 = time.perf_counter()
        model = P.load_model(src)
        t1 = time.perf_counter()
        out = quantize_model(model, make())
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        P.save_model(out, dst, external_data=os.path.basename(dst) + ".data")
        t3 = time.perf_counter()
        calls = {}
        for n in out.graph.node:
            if n.domain == "quant" or n.op_type == "MatMulNBits":
                calls[n.op_type] = calls.get(n.op_type, 0) + 1
        logits = GraphRunner(P.load_model(dst), outputs=["logits"], device="cuda")(feed)["logits"]
        rel = ((logits - float_logits).norm() / float_logits.norm()).item()
        report[name] = {"load_s": round(t1 - t0, 4), "quantize_s": round(t2 - t1, 3), "save_s": round(t3 - t2, 3), "calls": calls,
                        "lm_head_left_float": any(n.name == "/lm_head/MatMul" and n.op_type == "MatMul" for n in out.graph.node),
                        "result_bytes": os.path.getsize(dst) + os.path.getsize(dst + ".data"), "logits_rel_err": round(rel, 4)}
    if verbose:
        print(json.dumps(report))
    if not keep:
        shutil.rmtree(work, ignore_errors=True)
    return report


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--layers", type=int, default=18)
    ap.add_argument("--vocab", type=int, default=262144)
    ap.add_argument("--samples", type=int, default=64)
    ap.add_argument("--block", type=int, default=256)
    ap.add_argument("--dir", default=None)
    ap.add_argument("--static-samples", type=int, default=512, help="calibration sequences of the static int8 configuration (0: skip it)")
    a = ap.parse_args()
    main(a.layers, a.vocab, a.samples, a.block, a.dir, static_samples=a.static_samples)
