"""The `quant`-domain functions a quantized model calls, as FunctionProto messages (reference:
qfunctions/_qdq/qmatmul.py, _qdq/qgemm.py, _qlinear/qmatmul.py, _qlinear/qgemm.py, register.py:5-6).

The reference writes each function as an onnxscript script and registers its proto; the bodies are short compositions of
standard operators, and the fourteen QDQ variants differ only in which of {weights, bias, input, output} is wrapped in
Q -> DQ and whether the activation parameters are inputs (static) or come from DynamicQuantizeLinear.  Here ONE builder
composes them from those choices; the names, input orders and operator sequences are the reference's (the rewrite rules
pass arguments positionally, so the input order is part of the contract: qrules/_qdq/*.py).

  MatMul family  X, W, w_scale, w_zero_point [, x_scale, x_zero_point] [, out_scale, out_zero_point]
  Gemm family    X, W, B, w_scale, w_zero_point, b_scale, b_zero_point [, x_...] [, out_...]
  grouped        ... , original_transposed_shape; W [K, N] is transposed and reshaped to [N*K/g, g], dequantized blockwise
                 (block_size = g on the last axis) and brought back to [K, N]; 4-bit W / zero points are cast to INT8 first
                 because onnxruntime's Reshape does not take 4-bit tensors (qmatmul.py:192-194)
  QLinearMatMul / QLinearGemm   Q(X) -> QLinearMatMul | com.microsoft::QGemm -> DQ
"""
from __future__ import annotations

from .onnx_proto import DataType, Message, make_node

__all__ = ["QUANT_DOMAIN", "MS_DOMAIN", "FUNCTION_OPSET", "build_function", "function_names"]

QUANT_DOMAIN, QUANT_VERSION = "quant", 1             # qfunctions/register.py:5
MS_DOMAIN, MS_VERSION = "com.microsoft", 1           # qfunctions/register.py:6
FUNCTION_OPSET = 21                                  # opset.py: the bodies are written against opset 21

_MATMUL_QDQ = {(None, None): "QMatMulWeightsOnlyQDQ", ("static", None): "QMatMulWeightStaticInputQDQ",
               (None, "static"): "QMatMulWeightStaticOutputQDQ", ("static", "static"): "QMatMulWeightStaticInputOutputQDQ",
               ("dynamic", None): "QMatMulWeightDynamicInputQDQ", (None, "dynamic"): "QMatMulWeightDynamicOutputQDQ",
               ("dynamic", "dynamic"): "QMatMulWeightDynamicInputOutputQDQ"}
_GEMM_QDQ = {(None, None): "QGemmWeightsOnlyQDQ", ("static", None): "QGemmWeightInputQDQ",
             (None, "static"): "QGemmWeightOutputQDQ", ("static", "static"): "QGemmWeightInputOutputQDQ",
             ("dynamic", None): "QGemmWeightDynamicInputQDQ", (None, "dynamic"): "QGemmWeightDynamicOutputQDQ",
             ("dynamic", "dynamic"): "QGemmWeightDynamicInputOutputQDQ"}
_QDQ = {name: ("MatMul", modes) for modes, name in _MATMUL_QDQ.items()} | {name: ("Gemm", modes) for modes, name in _GEMM_QDQ.items()}


def function_names() -> list[str]:
    """Every function name the rules can emit (grouped ones exist per group size: `build_function(..., group_size=g)`)."""
    return sorted(_QDQ) + ["QGemmWeightsOnlyGrouped", "QMatMulWeightsOnlyGrouped", "QLinearGemm", "QLinearMatMul"]


def _function(name, inputs, outputs, nodes, doc, uses_ms=False) -> Message:
    opsets = [Message("OperatorSetIdProto", domain="", version=FUNCTION_OPSET)]
    if uses_ms:
        opsets.append(Message("OperatorSetIdProto", domain=MS_DOMAIN, version=MS_VERSION))
    return Message("FunctionProto", name=name, input=list(inputs), output=list(outputs), node=nodes, doc_string=doc,
                   opset_import=opsets, domain=QUANT_DOMAIN)


def _qdq_function(name: str) -> Message:
    op, (in_mode, out_mode) = _QDQ[name]
    gemm = op == "Gemm"
    inputs = ["X", "W"] + (["B"] if gemm else []) + ["w_scale", "w_zero_point"] + (["b_scale", "b_zero_point"] if gemm else [])
    nodes = [make_node("DequantizeLinear", ["W", "w_scale", "w_zero_point"], ["dequantized_weights"])]
    if gemm:
        nodes.append(make_node("DequantizeLinear", ["B", "b_scale", "b_zero_point"], ["dequantized_bias"]))
    x = "X"
    if in_mode == "static":
        inputs += ["x_scale", "x_zero_point"]
        nodes.append(make_node("QuantizeLinear", ["X", "x_scale", "x_zero_point"], ["x_quantized"]))
    elif in_mode == "dynamic":
        nodes.append(make_node("DynamicQuantizeLinear", ["X"], ["x_quantized", "x_scale", "x_zero_point"]))
    if in_mode:
        nodes.append(make_node("DequantizeLinear", ["x_quantized", "x_scale", "x_zero_point"], ["x_dequantized"]))
        x = "x_dequantized"
    nodes.append(make_node(op, [x, "dequantized_weights"] + (["dequantized_bias"] if gemm else []), ["out"]))
    result = "out"
    if out_mode == "static":
        inputs += ["out_scale", "out_zero_point"]
        nodes.append(make_node("QuantizeLinear", ["out", "out_scale", "out_zero_point"], ["out_quantized"]))
    elif out_mode == "dynamic":
        nodes.append(make_node("DynamicQuantizeLinear", ["out"], ["out_quantized", "out_scale", "out_zero_point"]))
    if out_mode:
        nodes.append(make_node("DequantizeLinear", ["out_quantized", "out_scale", "out_zero_point"], ["out_dequantized"]))
        result = "out_dequantized"
    what = {None: "", "static": "static ", "dynamic": "dynamic "}
    doc = (f"{op} on dequantized weights" + (" and bias" if gemm else "") +
           (f", {what[in_mode]}Q -> DQ on the input" if in_mode else "") +
           (f", {what[out_mode]}Q -> DQ on the output" if out_mode else "") + " (QDQ pattern).")
    return _function(name, inputs, [result], nodes, doc)


def _grouped_function(op: str, group_size: int, four_bit: bool) -> Message:
    gemm = op == "Gemm"
    name = "QGemmWeightsOnlyGrouped" if gemm else "QMatMulWeightsOnlyGrouped"
    inputs = ["X", "W"] + (["B"] if gemm else []) + ["w_scale", "w_zero_point"] + (["b_scale", "b_zero_point"] if gemm else []) + \
        ["original_transposed_shape"]
    nodes, w, zp = [], "W", "w_zero_point"
    if four_bit:
        nodes.append(make_node("Cast", ["W"], ["W_int8"], to=DataType.INT8))
        nodes.append(make_node("Cast", ["w_zero_point"], ["w_zero_point_int8"], to=DataType.INT8))
        w, zp = "W_int8", "w_zero_point_int8"
    nodes += [
        make_node("Transpose", [w], ["W_transposed"], perm=[1, 0]),
        make_node("Constant", [], ["grouped_shape"], value_ints=[-1, int(group_size)]),
        make_node("Reshape", ["W_transposed", "grouped_shape"], ["W_grouped"]),
        make_node("DequantizeLinear", ["W_grouped", "w_scale", zp], ["dequantized_groups"], block_size=int(group_size)),
        make_node("Reshape", ["dequantized_groups", "original_transposed_shape"], ["dequantized_transposed"]),
        make_node("Transpose", ["dequantized_transposed"], ["dequantized_weights"], perm=[1, 0]),
    ]
    if gemm:
        nodes.append(make_node("DequantizeLinear", ["B", "b_scale", "b_zero_point"], ["dequantized_bias"]))
    nodes.append(make_node(op, ["X", "dequantized_weights"] + (["dequantized_bias"] if gemm else []), ["out"]))
    doc = (f"{op} on weights dequantized in groups of {group_size} input channels"
           + (" (4-bit values cast to INT8 ahead of the reshape)" if four_bit else "") + ".")
    return _function(name, inputs, ["out"], nodes, doc)


def _qlinear_function(op: str) -> Message:
    if op == "MatMul":
        inputs = ["X", "W", "w_scale", "w_zero_point", "x_scale", "x_zero_point", "out_scale", "out_zero_point"]
        core = make_node("QLinearMatMul", ["x_quantized", "x_scale", "x_zero_point", "W", "w_scale", "w_zero_point",
                                           "out_scale", "out_zero_point"], ["out"])
        name = "QLinearMatMul"
    else:
        inputs = ["X", "W", "B", "w_scale", "w_zero_point", "x_scale", "x_zero_point", "out_scale", "out_zero_point"]
        core = make_node("QGemm", ["x_quantized", "x_scale", "x_zero_point", "W", "w_scale", "w_zero_point", "B",
                                   "out_scale", "out_zero_point"], ["out"], domain=MS_DOMAIN)
        name = "QLinearGemm"
    nodes = [make_node("QuantizeLinear", ["X", "x_scale", "x_zero_point"], ["x_quantized"]), core,
             make_node("DequantizeLinear", ["out", "out_scale", "out_zero_point"], ["out_dequantized"])]
    return _function(name, inputs, ["out_dequantized"], nodes, f"Q on the input, integer {op}, DQ on the output.",
                     uses_ms=op == "Gemm")


def build_function(name: str, *, group_size: int | None = None, four_bit: bool = False) -> Message:
    """The FunctionProto of `name` (`function_names()`); the grouped functions need the group size of the model being
    written and whether its weights are 4-bit (qmatmul.py:218-236: one function per group size under ONE name)."""
    if name in _QDQ:
        return _qdq_function(name)
    if name in ("QMatMulWeightsOnlyGrouped", "QGemmWeightsOnlyGrouped"):
        if not group_size or group_size <= 0:
            raise ValueError(f"{name} needs the group size")
        return _grouped_function("Gemm" if name.startswith("QGemm") else "MatMul", group_size, four_bit)
    if name == "QLinearMatMul":
        return _qlinear_function("MatMul")
    if name == "QLinearGemm":
        return _qlinear_function("Gemm")
    raise KeyError(f"no quant-domain function named '{name}'")
