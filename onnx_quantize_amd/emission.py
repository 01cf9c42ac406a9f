"""The emission contract of the rewrite rules as a plan (reference: qrules/base.py:15-81,
qrules/_qdq/matmul_to_qmatmul.py, _qdq/gemm_to_qgemm.py, _qlinear/matmul_to_qmatmul.py, _qlinear/gemm_to_qgemm.py,
qfunctions/factory.py, qfunctions/_qdq/qmatmul.py:218-270, _qdq/qgemm.py:267-316).

What a rule of the reference does to one MatMul / Gemm node is: (1) decide which branch applies (weights only /
static / dynamic, MatMulNBits or not), (2) call the numeric seam, (3) register initializers under fixed names and (4)
emit ONE call -- a `quant`-domain function or `com.microsoft::MatMulNBits` -- with a fixed input order and attributes.
(1), (3) and (4) are pure bookkeeping; this module restates them without `onnxscript` so that the numeric results of the
device path can be checked against everything the reference would have put into the graph for that node:
`plan_node(...)` returns the initializers (name, array) in registration order and the call (name, domain, version,
inputs, attributes).  tests/golden/emit.* hold what the reference's own rule methods recorded for 14 rule paths
(make_golden.py::gen_emit); tests/test_emission.py compares plan and recording entry by entry.

Building the ModelProto itself (pattern matching, function protos, serialisation) stays with the reference's
`quantize()` (integration.py); a graph writer can consume a plan as is.
"""
from __future__ import annotations

import dataclasses

import numpy as np

from .config import QConfig, QFormat, QuantizationStrategy, QWeightArgs
from .wire_format import _resolve_group_size, is_matmul_nbits_compatible

__all__ = ["EmissionPlan", "plan_node", "qfunction_name", "rule_for"]

QUANT_DOMAIN, QUANT_VERSION = "quant", 1             # qfunctions/register.py:5
MS_DOMAIN, MS_VERSION = "com.microsoft", 1           # qfunctions/register.py:6


@dataclasses.dataclass
class EmissionPlan:
    initializers: list                    # [(name, np.ndarray)] in the order `op.initializer` is called
    call: dict                            # {"name", "inputs": [value names | None], "attrs": {...}, "domain", "version"}
    # {initializer name: ONNX element type} where the NumPy dtype does not say it: 4-bit values travel in int8 / uint8
    # containers here (no ml_dtypes in the image), `ir.tensor` of the reference's ml_dtypes arrays makes INT4 / UINT4 tensors
    onnx_types: dict = dataclasses.field(default_factory=dict)


def _qdq_name(prefix: str, qconfig: QConfig) -> str:
    """qfunctions/_qdq/qmatmul.py:218-270 / qgemm.py:267-316 (prefix "QMatMul" / "QGemm")."""
    has_in, has_out = qconfig.input_activations is not None, qconfig.output_activations is not None
    is_static = (has_in and qconfig.input_activations.is_static) or (has_out and qconfig.output_activations.is_static)
    if qconfig.weights.strategy == QuantizationStrategy.GROUP:
        return f"{prefix}WeightsOnlyGrouped"                       # one function per group size, 4-bit variant same name
    matmul = prefix == "QMatMul"
    if is_static:
        if has_in and has_out:
            return "QMatMulWeightStaticInputOutputQDQ" if matmul else "QGemmWeightInputOutputQDQ"
        if has_in:
            return "QMatMulWeightStaticInputQDQ" if matmul else "QGemmWeightInputQDQ"
        if has_out:
            return "QMatMulWeightStaticOutputQDQ" if matmul else "QGemmWeightOutputQDQ"
        return f"{prefix}WeightsOnlyQDQ"
    if has_in and has_out:
        return f"{prefix}WeightDynamicInputOutputQDQ"
    if has_in:
        return f"{prefix}WeightDynamicInputQDQ"
    if has_out:
        return f"{prefix}WeightDynamicOutputQDQ"
    return f"{prefix}WeightsOnlyQDQ"


def qfunction_name(op_type: str, qconfig: QConfig) -> str:
    """qfunctions/factory.py:22-35: the name of the function a rule emits for `op_type` ("MatMul" | "Gemm")."""
    assert isinstance(qconfig.format, QFormat)
    if qconfig.format == QFormat.QLINEAR:
        return {"Gemm": "QLinearGemm", "MatMul": "QLinearMatMul"}[op_type]
    return _qdq_name({"Gemm": "QGemm", "MatMul": "QMatMul"}[op_type], qconfig)


def rule_for(node_op_type: str, has_bias: bool, qconfig: QConfig) -> tuple[str, str]:
    """qrules/factory.py:13-44 + the rule lists at the end of each rule module: (rule class name, the `op_type` property
    that rule hands to the function factory).  A Gemm WITHOUT bias is rewritten by a subclass of the MatMul rule that
    keeps its `op_type`, so it emits the MatMul function family (gemm_to_qgemm.py:9-25)."""
    qdq = qconfig.format == QFormat.QDQ
    if node_op_type == "MatMul":
        return ("MatMulToQMatMul" if qdq else "MatMulToQLinearMatMul"), "MatMul"
    if node_op_type == "Gemm":
        if has_bias:
            return ("GemmBiasToQGemmBias" if qdq else "GemmBiasToQLinearGemmBias"), "Gemm"
        return ("GemmToQGemm" if qdq else "GemmToQLinearGemm"), "MatMul"
    raise ValueError(f"no rewrite rule for op type {node_op_type!r}")


class _Value:
    def __init__(self, name, array):
        self.name = name
        self.const_value = None if array is None else _Tensor(array)


class _Tensor:
    def __init__(self, a):
        self._a = np.asarray(a)

    def numpy(self):
        return self._a


def plan_node(node_op_type: str, x_name: str, w, out_name: str, qconfig: QConfig, meta=None, bias=None, out=None,
              weight_arrays=None, quantize_bias=None) -> EmissionPlan:
    """`_plan_node` + the ONNX element types of the initializers whose NumPy container is wider than the type."""
    types: dict = {}
    plan = _plan_node(node_op_type, x_name, w, out_name, qconfig, meta, bias, out, weight_arrays, quantize_bias, types)
    plan.onnx_types = types
    return plan


def _plan_node(node_op_type, x_name, w, out_name, qconfig, meta, bias, out, weight_arrays, quantize_bias, types) -> EmissionPlan:
    """What `QRewriter._rewrite` (qrules/base.py:51-81) emits for one node.

    ``w`` / ``bias``: values with ``.name`` and ``.const_value.numpy()``; ``meta``: the node's calibration results
    (``input_scale`` ... , qrules/base.py:34-40); ``out``: the value whose producer carries ``meta["input"]`` for GPTQ.
    ``weight_arrays`` defaults to the device-resident seam (`seam.weight_arrays`), ``quantize_bias`` to
    `algorithms.rtn._quantize_bias`."""
    if weight_arrays is None:
        from .seam import weight_arrays
    if quantize_bias is None:
        from .algorithms.rtn import _quantize_bias as quantize_bias
    meta = meta or {}
    qconfig = QConfig(**qconfig.model_dump())                                     # base.py:57 works on its own copy
    _, fn_op_type = rule_for(node_op_type, bias is not None, qconfig)
    weights_only = qconfig.input_activations is None and qconfig.output_activations is None
    static_in = qconfig.input_activations is not None and qconfig.input_activations.is_static
    static_out = qconfig.output_activations is not None and qconfig.output_activations.is_static
    qconfig.weights.group_size = _resolve_group_size(w.const_value.numpy().shape[0], qconfig.weights.group_size, w.name)   # base.py:72
    qdq = qconfig.format == QFormat.QDQ
    inits: list = []

    def initializer(name, array):
        inits.append((name, np.asarray(array)))
        return _Value(name, array)

    def quantized(value, cfg, with_out, nbits=False):                             # qrules/_common.py:126-142
        q, s, z = weight_arrays(value, cfg, out if with_out else None, nbits)
        if not nbits and cfg.weights.dtype.bitwidth == 4:                         # `ir.tensor(w_q)` of an ml_dtypes array
            types[value.name] = int(cfg.weights.dtype.value)
            if np.asarray(z).dtype.kind in "iu":
                types[f"{value.name}/zero_point"] = int(cfg.weights.dtype.value)
        return (initializer(value.name, q), initializer(f"{value.name}/scale", s), initializer(f"{value.name}/zero_point", z))

    def act_qparams(kind, aargs):                                                 # qrules/base.py:15-40
        if aargs is None or not aargs.is_static:
            return None, None
        return (initializer(f"{out_name}/{kind}/scale", meta[f"{kind}_scale"]),
                initializer(f"{out_name}/{kind}/zero_point", meta[f"{kind}_zero_point"]))

    def qdq_bias():                                                               # _qdq/gemm_to_qgemm.py:47-62
        a = qconfig.weights
        cfg = QConfig(weights=QWeightArgs(dtype=a.dtype, is_symmetric=a.symmetric, strategy=QuantizationStrategy.TENSOR,
                                          scale_type=a.scale_dtype, clip_ratio=a.clip_ratio, mse=a.mse, reduce_range=a.reduce_range))
        return quantized(bias, cfg, False)

    names = lambda vals: [None if v is None else v.name for v in vals]            # noqa: E731
    quant = dict(domain=QUANT_DOMAIN, version=QUANT_VERSION)

    if not qdq:                                                                   # _qlinear/*.py: static only
        assert qconfig.format == QFormat.QLINEAR
        w_q, w_s, w_z = quantized(w, qconfig, True)
        i_s, i_z = act_qparams("input", qconfig.input_activations)
        o_s, o_z = act_qparams("output", qconfig.output_activations)
        args = [_Value(x_name, None), w_q]
        if bias is not None:                                                      # _qlinear/gemm_to_qgemm.py:48-57
            b_q, _, _ = quantize_bias(bias.const_value.numpy(), i_s.const_value.numpy(), w_s.const_value.numpy())
            args.append(initializer(bias.name, b_q))
        args += [w_s, w_z, i_s, i_z, o_s, o_z]
        return EmissionPlan(inits, dict(name=qfunction_name(fn_op_type, qconfig), inputs=names(args), attrs={}, **quant))

    if weights_only:
        if is_matmul_nbits_compatible(qconfig, w.name):                           # _qdq/matmul_to_qmatmul.py:58-82
            w_q, w_s, w_z = quantized(w, qconfig, True, nbits=True)
            k, n = w.const_value.numpy().shape
            args = [_Value(x_name, None), w_q, w_s, w_z, None] + ([bias] if bias is not None else [])
            return EmissionPlan(inits, dict(name="MatMulNBits", inputs=names(args), domain=MS_DOMAIN, version=MS_VERSION,
                                            attrs=dict(K=k, N=n, bits=qconfig.weights.dtype.bitwidth, block_size=qconfig.weights.group_size)))
        w_q, w_s, w_z = quantized(w, qconfig, True)                               # ..._weights_only_standard
        args = [_Value(x_name, None), w_q]
        if bias is not None:
            b_q, b_s, b_z = qdq_bias()
            args += [b_q, w_s, w_z, b_s, b_z]
        else:
            args += [w_s, w_z]
        if qconfig.weights.strategy == QuantizationStrategy.GROUP:
            shape = np.asarray(w.const_value.numpy().T.shape, dtype=np.int64)
            args.append(initializer(f"{w.name}/original_transposed_shape", shape))
        return EmissionPlan(inits, dict(name=qfunction_name(fn_op_type, qconfig), inputs=names(args),
                                        attrs=dict(num_bits=qconfig.weights.dtype.bitwidth), **quant))

    # static / dynamic QDQ.  The bias-less MatMul rule calls the seam WITHOUT `out` here (matmul_to_qmatmul.py:93,108)
    with_out = bias is not None
    w_q, w_s, w_z = quantized(w, qconfig, with_out)
    args = [_Value(x_name, None), w_q]
    if bias is not None:
        b_q, b_s, b_z = qdq_bias()
        args += [b_q, w_s, w_z, b_s, b_z]
    else:
        args += [w_s, w_z]
    if static_in or static_out:
        i_s, i_z = act_qparams("input", qconfig.input_activations)
        o_s, o_z = act_qparams("output", qconfig.output_activations)
        if i_s is not None:
            args += [i_s, i_z]
        if o_s is not None:
            args += [o_s, o_z]
    return EmissionPlan(inits, dict(name=qfunction_name(fn_op_type, qconfig), inputs=names(args), attrs={}, **quant))
