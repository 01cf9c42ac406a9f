"""`quantize()` on an ONNX file without the `onnx` / `onnx_ir` / `onnxscript` packages.

The reference's pipeline (quantize.py:28-80, pre_passes/__init__.py:47-98) is graph surgery around ONE numeric call per
node.  With the model held as parsed protobuf messages (`onnx_proto`), the surgery that a MatMul / Gemm model needs is
small enough to restate, and it lets a GPU box quantize a file end to end with nothing but this package:

  reference step                                         here
  ------------------------------------------------------ --------------------------------------------------------------
  ir.from_proto (a copy)                                  `Message.copy()` of the parsed model
  onnxscript.optimizer.optimize                           the part that decides which weights are constant: Identity
                                                          elimination, `Constant` nodes in weight / bias slots lifted, constant
                                                          folding with the folder's size limits (Transpose always); its other
                                                          rewrites are not restated
  version_converter.convert_version(target 21)            `_raise_opset`: the adapters a MatMul / Gemm export needs
                                                          (Reduce* axes, Split num_outputs), anything else refused by name
  NameFixPass                                             `_name_nodes`: unnamed nodes get names (the ignore patterns and
                                                          the activation initializers key on names)
  DuplicateInitializersPass (duplicate_initializer.py)    `_duplicate_shared_initializers`
  matmul_add_to_gemm_rule (onnxscript)                    `_fuse_matmul_add`: rank-2 operands only, like the rule
  StandarizeGemm[Bias] (standarize_gemm.py)               `_standardize_gemm`
  calibrate_model (calibrate.py:310-380)                  `_calibrate`: `GraphRunner` on the GPU -> `ActivationStream`
  _add_qconfig_to_nodes / get_target_nodes                `_target_nodes`
  preprocessors: AwqPass, SmoothQuantPass                 `_preprocess`: the searches on the GPU (`DeviceSearches`), the Mul /
                                                          initializer surgery here; then the post-calibration
  rewrite(model, get_qrules(qconfig))                     `emission.plan_node` per node -> initializers + one call
  model.functions.update / RemoveUnusedFunctionsPass      the functions the calls use (`onnx_functions`)
  DeduplicateInitializersPass(size_limit=1e9)             `_deduplicate_initializers`
  ir.to_proto                                             `onnx_proto.serialize`

Every number in the emitted model comes from the device path (`seam.weight_arrays`, `calibration_driver`); `weight_arrays`
can be injected, which is how the CPU suite drives the bookkeeping with the oracle.
"""
from __future__ import annotations

import logging
import os
import re

import numpy as np

from .config import QConfig
from .emission import plan_node
from .wire_format import _resolve_group_size
from .onnx_functions import FUNCTION_OPSET, MS_DOMAIN, QUANT_DOMAIN, build_function
from .onnx_proto import (DataType, Message, attribute_value, check_model, load_model, make_attribute, make_node, numpy_to_tensor, parse_model,
                         save_model, tensor_to_numpy)

__all__ = ["quantize_model", "quantize_model_sharded", "quantize_file", "apply_pre_passes", "target_nodes", "as_model"]

logger = logging.getLogger("onnx_quantize")

# AWQ / SmoothQuant: how much of the tapped activations a calibration walk holds before it folds them into running statistics
# (`ActivationStream.statistics_after_bytes`); a small model stays on the reference's arrays, a 7B-class model could not hold them
STATISTICS_AFTER_BYTES = 8 << 30
_MIN_SOURCE_OPSET = 13
# operators whose node form changed between opset 13 and 21 in a way this module does not adapt (version_converter would):
# GroupNormalization's scale and bias went from one value per group to one per channel, which needs the channel count
_NOT_ADAPTED = {"GroupNormalization": 21}
_GRID_SAMPLE_MODES = {"bilinear": "linear", "bicubic": "cubic"}                   # opset 20 renamed them
_REDUCE_AXES_TO_INPUT = {"ReduceMean", "ReduceMax", "ReduceMin", "ReduceProd", "ReduceL1", "ReduceL2", "ReduceLogSum",
                         "ReduceLogSumExp", "ReduceSumSquare"}                     # opset 18: `axes` moved from attribute to input


def as_model(model) -> Message:
    """bytes / path / parsed ModelProto -> a parsed ModelProto the caller's object is not aliased with."""
    if isinstance(model, Message):
        if model._type != "ModelProto":
            raise TypeError(f"model must be a ModelProto message, got {model._type}")
        return model.copy()
    if isinstance(model, (bytes, bytearray, memoryview)):
        return parse_model(bytes(model))
    if isinstance(model, (str, os.PathLike)):
        return load_model(model)
    raise TypeError(f"model must be ONNX bytes, a path or a parsed ModelProto (onnx_proto.parse_model), got {type(model)}")


# ------------------------------------------------------------------------------------------------------------ graph helpers
class _Const:
    """What the emission code reads of a constant value: `.name`, `.const_value.numpy()`; `device_value`: the same weight in
    HBM when the calibration walk left it there (the seam then skips its upload)."""

    def __init__(self, name, array, device_value=None, placeholder=False):
        self.name, self._a, self.device_value = name, array, device_value
        self.placeholder = placeholder       # `array` carries the shape only (zeros): the values live in `device_value` and nowhere else
        self.const_value = self

    def numpy(self):
        return self._a


class _Graph:
    """Name-indexed view of a GraphProto that keeps the message lists in step."""

    def __init__(self, graph: Message):
        self.g = graph
        self.inits = {t.name: t for t in graph.initializer}
        self.device_values: dict = {}                     # initializer name -> the tensor the calibration walk uploaded (read-only)
        # initializer name -> device tensor that IS the initializer's current value while its TensorProto still holds the old bytes
        # (weights AWQ / SmoothQuant rescaled in HBM: the seam quantizes the device copy, so the fp32 host copy -- 26 GB for a 7B
        # model -- is only made if something asks for it; nothing is serialised before `materialize()`)
        self.pending_host: dict = {}
        self.graph_inputs = {i.name for i in graph.input}
        self.graph_outputs = {o.name for o in graph.output}

    def array(self, name):
        self.materialize(name)
        t = self.inits.get(name)
        return None if t is None else tensor_to_numpy(t)

    def materialize(self, name=None) -> None:
        """Bring the put-off host copies up to date (one name, or all of them)."""
        for nm in ([name] if name is not None else list(self.pending_host)):
            value = self.pending_host.pop(nm, None)
            if value is not None:
                from .staging import download
                resident = self.device_values.get(nm)
                self.set_initializer(nm, download(value))
                if resident is not None:
                    self.device_values[nm] = resident

    def set_initializer(self, name, array, data_type=None):
        return self.set_tensor(numpy_to_tensor(name, array, data_type))

    def set_tensor(self, t):
        name = t.name
        self.device_values.pop(name, None)                # the copy in HBM, if any, is of the old contents
        self.pending_host.pop(name, None)
        if name in self.inits:
            old = self.inits[name]
            self.g.initializer[[id(x) for x in self.g.initializer].index(id(old))] = t
        else:
            self.g.initializer.append(t)
        self.inits[name] = t
        if name in self.graph_inputs:                     # IR < 4 files list initializers as inputs: keep the type in step
            for vi in self.g.input:
                if vi.name == name and vi.type is not None and vi.type.tensor_type is not None:
                    vi.type.tensor_type.elem_type = t.data_type
                    vi.type.tensor_type.shape = Message("TensorShapeProto", dim=[
                        Message("TensorShapeProto.Dimension", dim_value=int(d)) for d in t.dims])
        return t


def _attr(node, name, default=None):
    for a in node.attribute:
        if a.name == name:
            return attribute_value(a)
    return default


def _all_nodes(graph):
    for n in graph.node:
        yield n
        for a in n.attribute:
            if a.has("g"):
                yield from _all_nodes(a.g)
            for g in a.graphs:
                yield from _all_nodes(g)


# ------------------------------------------------------------------------------------------------------------ pre passes
def _raise_opset(model: Message, G: _Graph, target: int = FUNCTION_OPSET) -> None:
    """quantize.py:55 (`convert_version(model, target_version=op.version)`) for the operators exports of MatMul / Gemm models
    are made of.  Most of them did not change form between opset 13 and 21 (new optional inputs and attributes, wider type
    lists: Pad, Resize, Scatter*, Cast, Reshape ...: nothing to do); the ones that did are adapted the way onnx's converter
    does it -- Reduce* `axes` and DFT `axis` from attribute to input, Split `num_outputs`, RoiAlign's old default coordinate
    mode made explicit, GridSample's renamed modes, BatchNormalization with its single inference output --; an operator
    from `_NOT_ADAPTED` that crosses its change is refused by name rather than silently mis-declared."""
    default = [o for o in model.opset_import if not o.domain or o.domain == "ai.onnx"]
    if not default:
        model.opset_import.append(Message("OperatorSetIdProto", domain="", version=target))
        return
    current = max(int(o.version or 1) for o in default)
    if current >= target:
        return
    if current < _MIN_SOURCE_OPSET:
        raise NotImplementedError(f"the model imports opset {current}; this writer raises opsets >= {_MIN_SOURCE_OPSET} to {target} "
                                  "(older models: run onnx's version converter first)")
    taken_axes_names: set = set()
    for n in _all_nodes(model.graph):
        if n.domain not in (None, "", "ai.onnx"):
            continue
        changed = _NOT_ADAPTED.get(n.op_type)
        if changed is not None and current < changed <= target:
            raise NotImplementedError(f"node '{n.name}': operator {n.op_type} changed in opset {changed} and this writer has no adapter "
                                      f"for it (model at opset {current}; run onnx's version converter to {target} first)")
        if n.op_type in _REDUCE_AXES_TO_INPUT and current < 18:
            axes = _attr(n, "axes")
            if axes is not None:
                name, k = f"{n.output[0]}/axes", 0
                while name in G.inits or name in taken_axes_names:      # If branches reuse output names; an initializer of that name may exist
                    k += 1
                    name = f"{n.output[0]}/axes_{k}"
                taken_axes_names.add(name)
                G.set_initializer(name, np.asarray(axes, dtype=np.int64))
                n.input = list(n.input)[:1] + [name]
                n.attribute = [a for a in n.attribute if a.name != "axes"]
        if n.op_type == "Split" and current < 18 and len(n.input) < 2 and _attr(n, "num_outputs") is None:
            n.attribute = list(n.attribute) + [make_attribute("num_outputs", len(n.output))]
        if n.op_type == "BatchNormalization" and current < 14 and len([o for o in n.output if o]) > 1:
            raise NotImplementedError(f"node '{n.name}': BatchNormalization with training outputs changed in opset 14 and this writer "
                                      f"has no adapter for it (model at opset {current})")
        if n.op_type == "RoiAlign" and current < 16 and _attr(n, "coordinate_transformation_mode") is None:
            n.attribute = list(n.attribute) + [make_attribute("coordinate_transformation_mode", "output_half_pixel")]
        if n.op_type == "GridSample" and current < 20 and _attr(n, "mode") in _GRID_SAMPLE_MODES:
            n.attribute = [a for a in n.attribute if a.name != "mode"] + [make_attribute("mode", _GRID_SAMPLE_MODES[_attr(n, "mode")])]
        if n.op_type == "DFT" and current < 20:
            name = f"{n.output[0]}/axis"
            G.set_initializer(name, np.asarray(_attr(n, "axis", 1), dtype=np.int64))
            n.input = (list(n.input) + [""])[:2] + [name]
            n.attribute = [a for a in n.attribute if a.name != "axis"]
    for o in default:
        o.version = target
    if (model.ir_version or 0) < 10:                      # 4-bit tensors and opset 21 belong to IR version 10
        model.ir_version = 10


def _name_nodes(G: _Graph) -> None:
    taken = {n.name for n in G.g.node if n.name}
    for i, n in enumerate(G.g.node):
        if not n.name:
            name = f"node_{n.op_type}_{i}"
            while name in taken:
                name += "_"
            n.name = name
            taken.add(name)


# what `onnxscript.optimizer.optimize` (quantize.py:52) does to constants, as far as it decides WHICH nodes have a constant weight
# or bias: Identity nodes are removed, and a node whose inputs are all constants is replaced by its value -- always for
# Transpose, otherwise only when every input has at most 8192 elements and the result at most 512 * 512 (the folder's defaults)
_FOLD_INPUT_LIMIT, _FOLD_OUTPUT_LIMIT = 8192, 512 * 512
_FOLD_ALWAYS = {"Transpose"}
_FOLDABLE = {"Transpose", "Reshape", "Cast", "Unsqueeze", "Squeeze", "Concat", "Slice", "Gather", "Add", "Sub", "Mul", "Div", "Neg", "Sqrt",
             "Pow", "Flatten", "Expand", "Where", "Equal", "Not", "Shape", "Tile", "Abs", "Exp", "Log", "Reciprocal", "ReduceSum", "ReduceMean",
             "ReduceMax", "ReduceMin"}


def _eliminate_identities(G: _Graph) -> None:
    """`y = Identity(x)`: the readers of y read x (exporters write these for tensors they found to be equal, e.g. the zero
    bias of one layer as the Identity of another's).  An Identity that produces a graph output stays."""
    rename, drop = {}, set()
    for n in G.g.node:
        if n.op_type == "Identity" and not n.domain and len(n.input) == 1 and n.output and n.output[0] not in G.graph_outputs:
            rename[n.output[0]] = rename.get(n.input[0], n.input[0])
            drop.add(id(n))
    if not drop:
        return
    G.g.node = [n for n in G.g.node if id(n) not in drop]
    for n in _all_nodes(G.g):
        if any(v in rename for v in n.input):
            n.input = [rename.get(v, v) for v in n.input]


def _fold_constants(model: Message, G: _Graph) -> None:
    from .graph_runner import GraphRunner, UnsupportedOperator
    default = [int(o.version or 1) for o in model.opset_import if not o.domain or o.domain == "ai.onnx"]
    evaluator = GraphRunner.evaluator(max(default or [1]))
    keep = []
    for n in G.g.node:
        foldable = (not n.domain and n.op_type in _FOLDABLE and n.input and all(v in G.inits for v in n.input if v)
                    and not any(a.has("g") or a.graphs for a in n.attribute) and not any(o in G.graph_outputs for o in n.output))
        if foldable and n.op_type not in _FOLD_ALWAYS:
            foldable = all(int(np.prod(G.inits[v].dims, dtype=np.int64)) <= _FOLD_INPUT_LIMIT for v in n.input if v)
        if not foldable:
            keep.append(n)
            continue
        try:
            values = evaluator.evaluate(n, [G.array(v) if v else None for v in n.input])
        except (UnsupportedOperator, ValueError, RuntimeError, TypeError):
            keep.append(n)
            continue
        if n.op_type not in _FOLD_ALWAYS and any(v is not None and v.size > _FOLD_OUTPUT_LIMIT for v in values):
            keep.append(n)
            continue
        for name, value in zip(n.output, values):
            if name and value is not None:
                G.set_initializer(name, np.ascontiguousarray(value))
    G.g.node = keep


def _lift_constant_weights(G: _Graph) -> None:
    """`ir.convenience.get_const_tensor` (the constant check of every rule and of `get_target_nodes`, calibrate.py:75-85) also
    sees a value that a `Constant` node produces.  Such a node feeding the weight or bias slot of a MatMul / Gemm becomes an
    initializer of the same name here, so that everything downstream deals with initializers only."""
    wanted = {v for n in G.g.node if n.op_type in ("MatMul", "Gemm") and not n.domain for v in list(n.input)[1:3] if v}
    drop = set()
    for n in G.g.node:
        if n.op_type != "Constant" or n.domain or not n.output or n.output[0] not in wanted:
            continue
        value = next((a for a in n.attribute if a.name == "value" and a.has("t")), None)
        if value is None or n.output[0] in G.inits:
            continue
        t = value.t.copy()
        t.name = n.output[0]
        G.g.initializer.append(t)
        G.inits[t.name] = t
        drop.add(id(n))
    if drop:
        G.g.node = [n for n in G.g.node if id(n) not in drop]


def _duplicate_shared_initializers(G: _Graph) -> None:
    """duplicate_initializer.py:38-68: an initializer read in more than one place keeps its first use; every further use gets
    a copy named `<name>_<i>` (tied weights: each consumer is then quantized on its own)."""
    all_uses: dict = {}
    for n in G.g.node:                                     # `ir.Value.uses()` of every top-level value, in graph order
        for i, v in enumerate(n.input):
            if v in G.inits:
                all_uses.setdefault(v, []).append((n, i))
    for name in list(G.inits):
        if name in G.graph_inputs or name in G.graph_outputs:
            continue
        uses = all_uses.get(name, ())
        if len(uses) <= 1:
            continue
        src = G.inits[name]
        for i, (node, idx) in enumerate(uses[1:], start=1):
            new = src.copy()
            new.name = f"{name}_{i}"
            G.g.initializer.append(new)
            G.inits[new.name] = new
            ins = list(node.input)
            ins[idx] = new.name
            node.input = ins


_SAME_RANK = {"Relu", "Tanh", "Sigmoid", "Erf", "Sqrt", "Exp", "Log", "Neg", "Abs", "Cast", "Identity", "Dropout", "Softmax",
              "LogSoftmax", "LayerNormalization", "Transpose", "LeakyRelu", "Gelu", "Clip", "Softplus", "Reciprocal", "Floor",
              "Ceil", "Round", "Not", "Sin", "Cos", "Trilu", "Concat", "Slice", "Tile"}
_BROADCAST = {"Add", "Sub", "Mul", "Div", "Pow", "Min", "Max", "Sum", "Where", "Equal", "Less", "Greater", "LessOrEqual",
              "GreaterOrEqual", "And", "Or"}


def _infer_ranks(G: _Graph) -> dict:
    """Ranks of the values a MatMul / Add chain can be fed from, by one walk in graph order (the reference has them from the
    shape inference inside `onnxscript.optimizer.optimize`): declared types, then the rank rules of the common operators;
    a value this does not cover has no entry and the rule that asks for it does not fire."""
    ranks: dict = {name: len(t.dims) for name, t in G.inits.items()}
    for vi in list(G.g.input) + list(G.g.value_info) + list(G.g.output):
        tt = vi.type.tensor_type if vi.type is not None else None
        if tt is not None and tt.shape is not None:
            ranks.setdefault(vi.name, len(tt.shape.dim))
    for n in G.g.node:
        if n.domain or not n.output or n.output[0] in ranks:
            continue
        ins = [ranks.get(v) for v in n.input if v]
        r = None
        if n.op_type == "Gemm":
            r = 2
        elif n.op_type == "MatMul" and len(ins) == 2 and None not in ins:
            a, b = ins
            r = 0 if (a == 1 and b == 1) else (b - 1 if a == 1 else (a - 1 if b == 1 else max(a, b)))
        elif n.op_type in _SAME_RANK and ins and ins[0] is not None:
            r = ins[0]
        elif n.op_type in _BROADCAST and ins and None not in ins:
            r = max(ins)
        elif n.op_type == "Flatten":
            r = 2
        elif n.op_type == "Reshape" and len(n.input) > 1 and n.input[1] in G.inits:
            r = int(G.inits[n.input[1]].dims[0]) if G.inits[n.input[1]].dims else None
        elif n.op_type in ("Unsqueeze", "Squeeze") and ins and ins[0] is not None:
            axes = _attr(n, "axes")
            if axes is None and len(n.input) > 1 and n.input[1] in G.inits:
                axes = G.array(n.input[1]).reshape(-1).tolist()
            if axes is not None:
                r = ins[0] + (len(axes) if n.op_type == "Unsqueeze" else -len(axes))
        elif n.op_type == "Gather" and len(ins) == 2 and None not in ins:
            r = ins[0] + ins[1] - 1
        if r is not None:
            ranks[n.output[0]] = r
    return ranks


def _fuse_matmul_add(G: _Graph) -> None:
    """onnxscript's `matmul_add_to_gemm_rule` (pre_passes/__init__.py:62): Add(MatMul(a, b), c) -> Gemm(a, b, c) when a and b
    are known to be matrices (rank 2) and the product feeds nothing else."""
    ranks = _infer_ranks(G)
    producer = {o: n for n in G.g.node for o in n.output if o}
    consumers: dict = {}
    for n in G.g.node:
        for v in n.input:
            if v:
                consumers.setdefault(v, []).append(n)
    drop = set()
    for add in G.g.node:
        if add.op_type != "Add" or add.domain or len(add.input) != 2:
            continue
        mm = producer.get(add.input[0])
        if mm is None or mm.op_type != "MatMul" or mm.domain or id(mm) in drop:
            continue
        if len(consumers.get(mm.output[0], [])) != 1 or mm.output[0] in G.graph_outputs:
            continue
        if ranks.get(mm.input[0]) != 2 or ranks.get(mm.input[1]) != 2:
            continue
        add.op_type = "Gemm"
        add.input = [mm.input[0], mm.input[1], add.input[1]]
        add.attribute = []
        drop.add(id(mm))
    if drop:
        G.g.node = [n for n in G.g.node if id(n) not in drop]


def _standardize_gemm(G: _Graph) -> None:
    """standarize_gemm.py:5-57: a Gemm whose weight is constant and whose `transB` is absent or non-zero is re-emitted as
    Gemm(x, w[, b], transB=0) with the weight transposed when it was `transB = 1`.  As in the reference the re-emitted node
    carries `transB` ONLY: `alpha`, `beta` and `transA` of the original are not copied (a Gemm with explicit `transB = 0` is
    left alone, attributes and all)."""
    for n in G.g.node:
        if n.op_type != "Gemm" or n.domain or len(n.input) < 2 or n.input[1] not in G.inits:
            continue
        trans_b = _attr(n, "transB")
        if trans_b == 0:
            continue
        if trans_b:
            w = G.array(n.input[1])
            G.set_initializer(n.input[1], np.ascontiguousarray(w.T))
        n.input = [v for v in list(n.input)[:3]]
        n.attribute = [make_attribute("transB", 0)]


def _target_nodes(G: _Graph, qconfig: QConfig) -> list:
    """calibrate.py:48-89 + the `check` methods of the rules (matmul_to_qmatmul.py:31-41, gemm_to_qgemm.py:13-44): op type
    selected, name not ignored, constant weight (and bias), Gemm with `transB = 0`.  Weights that are not matrices are left
    alone (the numeric path is defined on [K, N])."""
    ignores = [re.compile(p) for p in qconfig.ignore]
    out = []
    for n in G.g.node:
        if n.domain or n.op_type not in qconfig.target_op_types or len(n.input) < 2:
            continue
        if n.name and any(p.search(n.name) for p in ignores):
            continue
        if n.input[1] not in G.inits:
            continue
        if len(n.input) > 2 and n.input[2] and n.input[2] not in G.inits:
            continue
        if len(n.input) > 2 and not n.input[2]:
            continue
        if n.op_type == "Gemm" and _attr(n, "transB") != 0:
            continue
        if len(G.inits[n.input[1]].dims) != 2 or G.inits[n.input[1]].data_type != DataType.FLOAT:
            logger.warning("node '%s': weight '%s' is not a float32 matrix (the numeric path is defined on fp32 [K, N]); left as it is", n.name, n.input[1])
            continue
        out.append(n)
    return out


# ------------------------------------------------------------------------------------------------------------ calibration
def _needs_calibration(qconfig: QConfig) -> bool:
    """pre_passes/__init__.py:31-44."""
    for a in (qconfig.input_activations, qconfig.output_activations):
        if a is not None and a.is_static:
            return True
    if any(pre.requires_calibration for pre in qconfig.preprocessors):
        return True
    return bool(qconfig.weights and qconfig.weights.algorithm.requires_calibration)


def _model_inputs(G: _Graph):
    """[(name, shape with None / str for symbolic dimensions, NumPy dtype)] of the real inputs (initializers excluded)."""
    from .onnx_proto import _NP_OF
    out = []
    for vi in G.g.input:
        if vi.name in G.inits:
            continue
        tt = vi.type.tensor_type if vi.type is not None else None
        if tt is None or tt.shape is None:
            raise ValueError(f"model input '{vi.name}' has no tensor shape: calibration data must be given")
        shape = [int(d.dim_value) if d.dim_value is not None else (d.dim_param or None) for d in tt.shape.dim]
        out.append((vi.name, shape, np.dtype(_NP_OF[tt.elem_type])))
    return out


def _calibrate(model: Message, G: _Graph, targets, qconfig: QConfig, device, keep_inputs: bool = False) -> dict:
    """calibrate.py:310-380 with the activations consumed on the device, batch by batch (calibration_driver.py): returns
    {id(node): meta} with `input_scale` / `input_zero_point` / `output_scale` / `output_zero_point` (0-d arrays) and `input`:
    a `StreamedGptqInput` (the Hessian of the node's input) when only the weight algorithm needs the activations, or -- with
    `keep_inputs`, ahead of AWQ / SmoothQuant, which read and rescale the activations themselves -- the batches concatenated in
    HBM (calibrate.py:296-307), or, with `keep_inputs="statistics"` (the default of the device searches), what those searches
    need of them as running statistics (`ops.SearchStatistics`: Gram matrix, |x| sums and maxima) once the walk has tapped more
    than it should hold (`ActivationStream.statistics_after_bytes`, 8 GiB; below that the batches themselves).  Nodes that
    read the same value share ONE object either way, as they share one array in the reference."""
    from .calibration import get_calibrator
    from .calibration_driver import ActivationStream, generate_random_calibration_data, run_calibration
    from .graph_runner import GraphRunner
    from .reference_passes import StreamedGptqInput

    cal_in = qconfig.input_activations is not None and qconfig.input_activations.is_static
    cal_out = qconfig.output_activations is not None and qconfig.output_activations.is_static
    algo = qconfig.weights is not None and qconfig.weights.algorithm.requires_calibration
    in_names = [n.input[0] for n in targets]
    out_names = [n.output[0] for n in targets]
    pre = any(p.requires_calibration for p in qconfig.preprocessors)
    wanted = list(dict.fromkeys((in_names if (cal_in or algo or pre) else []) + (out_names if cal_out else [])))
    params = qconfig.calibration_params.model_dump()
    batch_size, num_samples = params.pop("batch_size"), params.pop("num_samples")
    params.pop("provider")                                 # the GPU this process owns runs the graph
    calibrator = get_calibrator(params.pop("method"), **params)
    inputs = _model_inputs(G)
    data = qconfig.calibration_data
    if data is None:
        data = generate_random_calibration_data(num_samples, inputs)
    # a model input that a target node reads directly is "produced" by the feed: the runner returns it like any other value
    # batches of one (small) shape replay a recorded pass; large products with constant weights take the fp16-piece GEMM
    # weights an earlier walk (or a rescale on the device) left in HBM are run on as they are: `G.device_values` is dropped for a
    # name whenever its initializer is set, so what it holds is current -- also where the TensorProto is not (`pending_host`)
    runner = GraphRunner(model, outputs=wanted, device=device, capture=True, matmul="pieces", constants=dict(G.device_values))
    stream = ActivationStream(calibrator=calibrator, input_names=in_names if cal_in else (), output_names=out_names if cal_out else (),
                              hessian_names=in_names if (algo and not keep_inputs) else (),
                              keep_names=in_names if keep_inputs is True else (),
                              statistics_names=in_names if keep_inputs == "statistics" else (), statistics_after_bytes=STATISTICS_AFTER_BYTES)
    run_calibration(runner, data, stream, num_samples=num_samples, batch_size=batch_size, input_names=[i[0] for i in inputs])
    G.device_values = {name: t for name, t in runner.constants.items() if t.is_cuda and t.ndim == 2}
    meta: dict = {id(n): {} for n in targets}
    for kind, on, names, aargs in (("input", cal_in, in_names, qconfig.input_activations),
                                   ("output", cal_out, out_names, qconfig.output_activations)):
        if not on:
            continue
        qparams = stream.input_qparams(aargs) if kind == "input" else stream.output_qparams(aargs)
        for n, name in zip(targets, names):
            if name in qparams:
                scale, zp = qparams[name]
                meta[id(n)][f"{kind}_scale"] = np.asarray(scale).astype(aargs.scale_dtype, copy=False)
                meta[id(n)][f"{kind}_zero_point"] = np.asarray(zp).astype(aargs.zp_dtype, copy=False)
    if keep_inputs == "statistics":
        for n, name in zip(targets, in_names):
            try:
                meta[id(n)]["input"] = stream.search_input(name)    # one object per value name: shared by its consumers
            except KeyError:
                pass                                                # a value the calibration data never reached
    elif keep_inputs:
        kept: dict = {}
        for n, name in zip(targets, in_names):
            if name not in kept:
                kept[name] = stream.kept(name)
            meta[id(n)]["input"] = kept[name]
    elif algo:
        shared: dict = {}
        for n, name in zip(targets, in_names):
            acc = stream.hessians.get(name)
            if acc is None:
                continue
            if name not in shared:
                shared[name] = StreamedGptqInput(name, acc.h, acc.n, (acc.n, acc.h.shape[0]))
            meta[id(n)]["input"] = shared[name]
    return meta


# ------------------------------------------------------------------------------------------------------------ AWQ / SmoothQuant
class DeviceSearches:
    """The numeric cores of the two pre-processing passes on the GPU (hip/ops.py: `oq_smooth_quant_scale_f32`,
    `oq_awq_scale_search_f32`, `oq_awq_clip_search_f32`).  `x`: the node's calibration input in HBM, `w`: the weight [K, N]
    as NumPy (uploaded once per node)."""

    def __init__(self):
        self._w = (None, None)

    def _dev(self, w):
        from .staging import upload
        if self._w[0] is not w:
            self._w = (w, upload(w))
        return self._w[1]

    def resident(self, w, tensor) -> None:
        """`tensor` is `w` in HBM already (same shape, fp32, contiguous): searched on as it is."""
        if tuple(tensor.shape) == tuple(w.shape) and tensor.is_cuda and tensor.is_contiguous() and str(tensor.dtype) == "torch.float32":
            self._w = (w, tensor)

    @staticmethod
    def _args(a):
        return a.dtype.key, a.strategy.value, a.group_size, bool(a.symmetric), bool(a.reduce_range)

    wants = "statistics"     # what `_calibrate` should leave in `meta["input"]`: running statistics instead of the activations

    def smooth_quant_scale(self, x, w, alpha):
        from .hip import ops
        from .staging import download
        fn = ops.smooth_quant_scale_stats if isinstance(x, ops.SearchStatistics) else ops.smooth_quant_scale
        return download(fn(x, self._dev(w), float(alpha)))

    def awq_scale_search(self, x, w, a):
        from .hip import ops
        from .staging import download
        fn = ops.awq_scale_search_stats if isinstance(x, ops.SearchStatistics) else ops.awq_scale_search
        best, _losses = fn(x, self._dev(w), *self._args(a))
        return download(best)

    def awq_clip_search(self, x, w, a):
        from .hip import ops
        fn = ops.awq_clip_search_stats if isinstance(x, ops.SearchStatistics) else ops.awq_clip_search
        ratio, _losses = fn(x, self._dev(w), *self._args(a))
        return float(ratio)

    def scale_rows(self, w, scale, want_host=True):
        """awq.py:187 / smooth_quant.py:114 (`scale.reshape(-1, 1) * weights`) on the copy of the weight the search already
        uploaded: (the product on the host, the same product in HBM for the seam -- no second upload, no pass over the weight in
        NumPy).  One fp32 multiply per element either way: the same bits."""
        import torch
        from .staging import download
        updated = self._dev(w) * torch.from_numpy(np.ascontiguousarray(scale, dtype=np.float32)).to(self._dev(w).device).reshape(-1, 1)
        host = download(updated) if want_host else _OnDevice(updated)       # (a stand-in the clip search is handed back)
        self._w = (host, updated)                             # the clip search that may follow runs on the updated weight
        return host, updated


class _OnDevice:
    """A rescaled weight that exists in HBM only (its host copy has been put off, `_Graph.pending_host`)."""

    def __init__(self, tensor):
        self.tensor = tensor


def _divide_in_place(x, scale: np.ndarray) -> None:
    """`node.meta["input"] /= scale.reshape((1, -1))` (smooth_quant.py:121, awq.py:191): IN PLACE on the array the nodes that
    read one value share -- the next consumer of that value (k after q, up after gate) searches on inputs already divided
    by its neighbour's scale.  That is the reference as written; the shared object keeps it so."""
    if isinstance(x, np.ndarray):
        x /= scale.reshape((1, -1))
        return
    import torch
    s = torch.from_numpy(np.ascontiguousarray(scale, dtype=np.float32))
    if hasattr(x, "abs_sum"):                               # ops.SearchStatistics: the same rescale on the running statistics
        x.divide(s)
        return
    x.div_(s.to(x.device).reshape(1, -1))


def _preprocess(G: _Graph, targets, qconfig: QConfig, meta: dict, searches, defer_host: bool = False) -> dict:
    """pre_passes/__init__.py:72-83: every preprocessor's pass over the target nodes in graph order.  Both passes do the same
    surgery around different searches (smooth_quant.py:91-134, awq.py:114-204): the scale is folded into the weight's rows,
    a `Mul` by 1 / scale goes in front of the node (initializer `<node output>_scale`), the node's calibration input is
    divided by the scale.  AWQ's clip search (awq.py:206-259) leaves a per-node `clip_ratio`.  Returns {id(node): QConfig}
    for the nodes whose configuration changed."""
    per_node: dict = {}
    if len([p for p in qconfig.preprocessors if p.preprocessing_type in ("awq", "smooth_quant")]) > 1:
        raise NotImplementedError("more than one rescaling preprocessor: both would register the initializer '<output>_scale'")
    for pre in qconfig.preprocessors:
        kind = pre.preprocessing_type
        if kind not in ("awq", "smooth_quant"):
            raise NotImplementedError(f"preprocessor '{kind}' has no restatement in this writer")
        for node in targets:
            node_meta = meta.get(id(node), {})
            if "input" not in node_meta:
                continue                                     # a node the calibration data never reached
            x = node_meta["input"]
            w_name, out_name = node.input[1], node.output[0]
            w = G.array(w_name)
            cfg = per_node.get(id(node), qconfig)
            if hasattr(searches, "resident") and G.device_values.get(w_name) is not None:
                searches.resident(w, G.device_values[w_name])           # the calibration walk left this weight in HBM
            if kind == "smooth_quant":
                scale = np.asarray(searches.smooth_quant_scale(x, w, pre.alpha), dtype=np.float32)
            else:
                scale = np.asarray(searches.awq_scale_search(x, w, cfg.weights), dtype=np.float32)
            resident = None
            if hasattr(searches, "scale_rows"):
                updated, resident = searches.scale_rows(w, scale, want_host=not defer_host)
            else:
                updated = np.multiply(scale.reshape(-1, 1), w)
            name = f"{out_name}_scale"
            if name in G.inits:
                raise ValueError(f"an initializer named '{name}' exists already")
            G.set_initializer(name, (1.0 / scale).astype(np.float32))
            _divide_in_place(x, scale)
            mul_out = f"{out_name}_scaled_input"
            mul = make_node("Mul", [node.input[0], name], [mul_out], name=f"{node.name}/scale_input")
            nodes = list(G.g.node)
            nodes.insert([id(n) for n in nodes].index(id(node)), mul)
            G.g.node = nodes
            node.input = [mul_out] + list(node.input)[1:]
            if isinstance(updated, _OnDevice):
                G.pending_host[w_name] = resident             # the TensorProto keeps the old bytes until somebody asks (`_Graph.array`)
            else:
                G.set_initializer(w_name, updated.astype(np.float32, copy=False))
            if resident is not None:
                G.device_values[w_name] = resident            # the seam quantizes this copy instead of uploading the weight again
            if kind == "awq" and pre.clip_search:
                ratio = searches.awq_clip_search(x, updated, cfg.weights)
                changed = cfg.model_copy()
                changed.weights = cfg.weights.model_copy()
                changed.weights.clip_ratio = ratio
                per_node[id(node)] = changed
    return per_node


# ------------------------------------------------------------------------------------------------------------ post passes
def _same_bytes(a, b) -> bool:
    """Content equality of two byte buffers, cheap when they differ: a few samples first, the whole only when those agree
    (weights of one shape are legion in a transformer and never equal; hashing them all would read the whole model)."""
    n = len(a)
    if n != len(b):
        return False
    if n > 4096:
        step = max(1, n // 64)
        for off in list(range(0, n - 64, step)) + [n - 64]:
            if a[off:off + 64] != b[off:off + 64]:
                return False
    return a == b


def _deduplicate_initializers(G: _Graph, size_limit: float = 1e9) -> None:
    """`DeduplicateInitializersPass(size_limit=1e9)` (quantize.py:75): initializers with the same element type, shape and
    bytes become one (the first in graph order stays); graph inputs / outputs and string tensors are left alone."""
    seen: dict = {}
    rename: dict = {}
    keep = []
    for t in G.g.initializer:
        if t.name in G.graph_inputs or t.name in G.graph_outputs or t.data_type == DataType.STRING or not t.has("raw_data") \
                or len(t.raw_data) > size_limit:
            keep.append(t)
            continue
        raw = t.raw_data if isinstance(t.raw_data, memoryview) else memoryview(t.raw_data)
        candidates = seen.setdefault((t.data_type, tuple(t.dims), len(raw)), [])
        first = next((name for name, other in candidates if _same_bytes(raw, other)), None)
        if first is None:
            candidates.append((t.name, raw))
            keep.append(t)
        else:
            rename[t.name] = first
    if not rename:
        return
    G.g.initializer = keep
    for n in _all_nodes(G.g):
        if any(v in rename for v in n.input):
            n.input = [rename.get(v, v) for v in n.input]
    G.inits = {t.name: t for t in keep}


def _remove_unused_initializers(G: _Graph) -> None:
    used = {v for n in _all_nodes(G.g) for v in n.input if v} | G.graph_outputs
    G.g.initializer = [t for t in G.g.initializer if t.name in used]
    G.inits = {t.name: t for t in G.g.initializer}


# ------------------------------------------------------------------------------------------------------------ the pipeline
class _PackedSeam:
    """The device seam as the writer's default provider: the same three arrays, except that 4-bit integers of a matrix come back
    already nibble-packed from the device (`seam.weight_arrays(..., packed4=True)`) and go into the TensorProto as they are --
    packing 6.5 G values in NumPy was a third of a 7B-model GPTQ run.  The emission code reads shapes and dtypes only, so it is
    handed a zero-stride placeholder of the right shape in their place."""

    def __init__(self):
        self.packed: dict = {}

    def __call__(self, value, cfg, out, nbits):
        from .seam import weight_arrays
        a = cfg.weights
        w = value.const_value.numpy()
        algorithm = getattr(a.algorithm, "algorithm_type", None)
        if nbits or a.dtype.bitwidth != 4 or w.ndim != 2 or algorithm not in ("rtn", "gptq", "hqq"):
            return weight_arrays(value, cfg, out, nbits)
        q, s, z = weight_arrays(value, cfg, out, nbits, packed4=True)
        self.packed[value.name] = (q, tuple(w.shape))
        return np.broadcast_to(np.zeros((), dtype=a.dtype.np_dtype), w.shape), s, z


class _Out:
    """What the seam reads of a node's output value: `out.producer().meta`."""

    def __init__(self, meta):
        self.meta = meta

    def producer(self):
        return self


def target_nodes(model, qconfig: QConfig) -> list:
    """Which nodes `quantize()` would rewrite, without touching a GPU: the structural pre-passes (opset, names, Identity / constant
    folding, duplicated initializers, MatMul + Add -> Gemm, Gemm with `transB = 0`) followed by the selection of
    `get_target_nodes` (calibrate.py:48-89) and the rules' checks.  Returns [(node name, op type, weight name, [K, N])] in graph
    order; the caller's model is not modified."""
    model = as_model(model)
    if model.graph is None:
        raise ValueError("the model has no graph")
    G = _structural_passes(model)
    return [(n.name, n.op_type, n.input[1], [int(d) for d in G.inits[n.input[1]].dims]) for n in _target_nodes(G, qconfig)]


def _structural_passes(model: Message) -> _Graph:
    """quantize.py:50-55 + the standard passes of pre_passes/__init__.py:58-65, in the reference's order."""
    G = _Graph(model.graph)
    _raise_opset(model, G)
    _name_nodes(G)
    _eliminate_identities(G)
    _lift_constant_weights(G)
    _fold_constants(model, G)
    _duplicate_shared_initializers(G)
    _fuse_matmul_add(G)
    _standardize_gemm(G)
    return G


class Prepared:
    """A model after the pre-passes: the graph view, the nodes to quantize, their calibration results and the per-node
    configurations a preprocessor changed (AWQ's clip ratio)."""

    def __init__(self, model, graph, targets, meta, per_node):
        self.model, self.graph, self.targets, self.meta, self.per_node = model, graph, targets, meta, per_node


def apply_pre_passes(model, qconfig: QConfig, *, device="cuda", calibrate=None, searches=None, post_calibration="if_read",
                     _defer_host_copies: bool = False) -> Prepared:
    """quantize.py:50-59 + pre_passes/__init__.py:47-98: opset, names, duplicated initializers, MatMul + Add -> Gemm, Gemm with
    `transB = 0`, calibration, the preprocessors' passes and the calibration after them.  `post_calibration`: "if_read" skips
    the second calibration walk when nothing downstream reads its results (weight-only RTN behind AWQ / SmoothQuant: no static
    activation, no Hessian); "always" runs it whenever a preprocessor asks for it, as the reference does."""
    model = as_model(model)
    if model.graph is None:
        raise ValueError("the model has no graph")
    G = _structural_passes(model)
    targets = _target_nodes(G, qconfig)
    if not targets:
        # an fp16 / bf16 export (the usual half-precision form) has constant MatMul / Gemm weights and not one fp32
        # matrix among them; returning it untouched with a success status would look like a quantized model
        halves = [n.input[1] for n in G.g.node if not n.domain and n.op_type in qconfig.target_op_types and len(n.input) > 1 and n.input[1] in G.inits
                  and len(G.inits[n.input[1]].dims) == 2 and G.inits[n.input[1]].data_type in (DataType.FLOAT16, DataType.BFLOAT16)]
        if halves:
            raise NotImplementedError(f"none of the model's {len(halves)} constant MatMul / Gemm weights is float32 (e.g. '{halves[0]}' is half precision): "
                                      "the numeric path is defined on fp32 [K, N] matrices -- convert the weights to float32 first")
    calibrate = calibrate or _calibrate
    per_node: dict = {}
    meta: dict = {}
    searches = searches or DeviceSearches()
    if _needs_calibration(qconfig) and targets:
        keep = (getattr(searches, "wants", True) if calibrate is _calibrate else True) if qconfig.preprocessors else False
        meta = calibrate(model, G, targets, qconfig, device, keep_inputs=keep)
    if qconfig.preprocessors and targets:                   # pre_passes/__init__.py:72-88
        per_node = _preprocess(G, targets, qconfig, meta, searches, defer_host=_defer_host_copies)
        _name_nodes(G)
        read = any(a is not None and a.is_static for a in (qconfig.input_activations, qconfig.output_activations)) or \
            bool(qconfig.weights.algorithm.requires_calibration)
        if any(p.requires_post_calibration for p in qconfig.preprocessors) and (read or post_calibration == "always"):
            logger.info("Re-calibrating the model after pre-processing...")
            if calibrate is not _calibrate:
                G.materialize()                              # a provider that reads the rescaled weights from the model's bytes
            meta = calibrate(model, G, targets, qconfig, device, keep_inputs=post_calibration == "always" and not read)
    qconfig.calibration_data = None                         # pre_passes/__init__.py:90: the caller's configuration lets go of the data
    if not _defer_host_copies:
        G.materialize()
    return Prepared(model, G, targets, meta, per_node)


def _on(device):
    """The kernels run on torch's CURRENT device (hip/ops.py refuses tensors of another one): `device="cuda:1"` makes that GPU
    current for the duration of the call."""
    import contextlib
    if isinstance(device, str) and not device.startswith("cuda"):
        return contextlib.nullcontext()
    import torch
    dev = torch.device(device)
    if dev.type != "cuda" or dev.index is None:
        return contextlib.nullcontext()
    return torch.cuda.device(dev.index)


def quantize_model(model, qconfig: QConfig, *, device="cuda", weight_arrays=None, quantize_bias=None, calibrate=None,
                   searches=None) -> Message:
    """quantize.py:28-80 on a parsed ModelProto / ONNX bytes / a path.  Returns a new parsed ModelProto (`onnx_proto.serialize`
    gives the file).  `weight_arrays` / `quantize_bias` / `calibrate` / `searches`: the numeric providers (default: the
    device-resident seam, the HIP bias kernel, the on-device calibration walk `_calibrate` and `DeviceSearches`; tests inject
    the oracle).  `calibrate(model, graph view, target nodes, qconfig, device, keep_inputs=False)` returns {id(node): meta}
    like `_calibrate`."""
    if not isinstance(qconfig, QConfig):
        raise TypeError(f"qconfig must be a QConfig, got {type(qconfig)}")
    if qconfig.weights is None and qconfig.input_activations is None and qconfig.output_activations is None:
        logger.info("No quantization parameters specified in qconfig. Returning original model.")
        return as_model(model)
    with _on(device):
        prepared = apply_pre_passes(model, qconfig, device=device, calibrate=calibrate, searches=searches, _defer_host_copies=weight_arrays is None)
        return _emit(prepared, qconfig, weight_arrays if weight_arrays is not None else _PackedSeam(), quantize_bias)


def _plan(prepared: Prepared, node, qconfig: QConfig, weight_arrays, quantize_bias, on_device: bool = False):
    G = prepared.graph
    has_bias = len(node.input) > 2
    name = node.input[1]
    resident = G.device_values.get(name)
    cfg = prepared.per_node.get(id(node), qconfig)
    dims = tuple(int(d) for d in G.inits[name].dims)
    if (on_device and name in G.pending_host and resident is not None and tuple(resident.shape) == dims
            and getattr(cfg.weights.algorithm, "algorithm_type", None) in ("rtn", "hqq", "gptq")):
        host = np.broadcast_to(np.float32(0), dims)           # the seam reads the shape only: it quantizes the copy in HBM
        placeholder = True
    else:
        host = G.array(name)
        placeholder = False
    w = _Const(name, host, resident, placeholder=placeholder)
    b = _Const(node.input[2], G.array(node.input[2])) if has_bias else None
    node_meta = prepared.meta.get(id(node), {})
    return plan_node(node.op_type, node.input[0], w, node.output[0], cfg, node_meta, bias=b,
                     out=_Out(node_meta), weight_arrays=weight_arrays, quantize_bias=quantize_bias)


def _emit(prepared: Prepared, qconfig: QConfig, weight_arrays, quantize_bias) -> Message:
    """quantize.py:60-78: the rules' rewrites node by node, the functions, the opset imports, the post passes."""
    model, G, targets = prepared.model, prepared.graph, prepared.targets

    algorithm = qconfig.weights.algorithm
    if getattr(algorithm, "algorithm_type", None) == "gptq" and isinstance(weight_arrays, _PackedSeam):
        from .reference_passes import StreamedGptqInput
        from .seam import prefactor_streamed

        streamed = [m["input"] for m in prepared.meta.values() if isinstance(m.get("input"), StreamedGptqInput)]
        if streamed:                                          # all inverse factors of the model in batched chains, ahead of the nodes
            prefactor_streamed(streamed, float(algorithm.percdamp), bool(algorithm.actorder))

    used_functions: dict = {}
    domains = set()
    readers: dict = {}
    for n in _all_nodes(G.g):
        for v in n.input:
            if v:
                readers[v] = readers.get(v, 0) + 1
    for node in targets:
        w_name = node.input[1]
        if readers[w_name] > 1 or w_name in G.graph_outputs:
            # an initializer that `_duplicate_shared_initializers` had to leave shared (it is a graph input or output,
            # duplicate_initializer.py:47-52): its integers would replace it under the other readers' feet
            raise ValueError(f"node '{node.name}': its weight '{w_name}' is a graph output or is read elsewhere and could not be duplicated "
                             "(it is listed among the graph's inputs / outputs); remove it from there or ignore this node")
        in_channels = int(G.inits[w_name].dims[0])
        plan = _plan(prepared, node, qconfig, weight_arrays, quantize_bias, on_device=isinstance(weight_arrays, _PackedSeam) or getattr(weight_arrays, "on_device", False))
        packed = getattr(weight_arrays, "packed", {})
        for name, array in plan.initializers:
            if name != w_name and name in G.inits and (readers.get(name, 0) > 1 or name in G.graph_outputs):
                raise ValueError(f"node '{node.name}': the initializer '{name}' it rewrites is a graph output or is read elsewhere and could "
                                 "not be duplicated; remove it from the graph's inputs / outputs or ignore this node")
            if name in packed:                                # 4-bit integers packed on the device (`_PackedSeam`)
                raw, shape = packed.pop(name)
                G.set_tensor(Message("TensorProto", dims=[int(d) for d in shape], data_type=int(plan.onnx_types[name]), name=name,
                                     raw_data=memoryview(np.ascontiguousarray(raw).reshape(-1)).cast("B")))
            else:
                G.set_initializer(name, np.asarray(array), plan.onnx_types.get(name))
        call = plan.call
        node.op_type, node.domain = call["name"], call["domain"]
        node.input = ["" if v is None else v for v in call["inputs"]]
        node.attribute = [make_attribute(k, v) for k, v in call["attrs"].items()]
        domains.add(call["domain"])
        if call["domain"] == QUANT_DOMAIN:
            group = None
            if call["name"].endswith("WeightsOnlyGrouped"):                       # base.py:72: the group size this weight resolved to
                group = int(_resolve_group_size(in_channels, qconfig.weights.group_size, w_name))
            used_functions.setdefault((call["name"], group), []).append(node)

    # qfunctions/__init__.py:11-22 + RemoveUnusedFunctionsPass: the functions the calls use, nothing else
    four_bit = qconfig.weights is not None and qconfig.weights.dtype.bitwidth == 4
    group_sizes = sorted({g for (_, g) in used_functions if g is not None})
    for (name, group), nodes in sorted(used_functions.items(), key=lambda kv: (kv[0][0], kv[0][1] or 0)):
        fn = build_function(name, group_size=group, four_bit=four_bit)
        if group is not None and len(group_sizes) > 1:
            # the reference registers every group size under ONE name and the last one wins (qmatmul.py:218-236 through
            # qfunctions/__init__.py:17-20), which leaves the other calls with the wrong block size; here each size is its own
            # overload of the name, so a model whose weights resolve to different group sizes stays correct
            fn.overload = f"g{group}"
            for n in nodes:
                n.overload = fn.overload
        model.functions.append(fn)
    have = {o.domain or "" for o in model.opset_import}
    if QUANT_DOMAIN in domains and QUANT_DOMAIN not in have:
        model.opset_import.append(Message("OperatorSetIdProto", domain=QUANT_DOMAIN, version=1))
    uses_ms = MS_DOMAIN in domains or any(n.domain == MS_DOMAIN for f in model.functions for n in f.node)
    if uses_ms and MS_DOMAIN not in have:
        model.opset_import.append(Message("OperatorSetIdProto", domain=MS_DOMAIN, version=1))

    G.materialize()                                         # (a rescaled weight no rewrite replaced: its host copy is due now)
    _remove_unused_initializers(G)
    _deduplicate_initializers(G)
    check_model(model)                                      # what leaves is structurally sound, or the caller hears why not
    return model


def quantize_model_sharded(model, qconfig: QConfig, *, group=None, device="cuda", weight_arrays=None, quantize_bias=None,
                           calibrate=None, searches=None):
    """`quantize_model` with the weights of the model spread over the ranks of a process group (BASELINE.json's
    last configuration): every rank parses the file (its tensors are memory-mapped, a rank only touches the weights it
    quantizes) and runs the pre-passes and the calibration walk for itself -- they decide WHAT is quantized and with which
    statistics, and replicating them needs no exchange --, then each rank runs the numeric seam for the nodes a cost-balanced
    plan gives it (`sharding.plan_lpt`: nodes that read one value stay together, so a Hessian is factored once), the three
    arrays of every weight are gathered on rank 0 (`sharding.quantize_sharded`: one gather at the end, RCCL over xGMI on a GPU
    node), and rank 0 emits the model.  Returns the model on rank 0, None on the other ranks.  Without an initialised process
    group it is `quantize_model`."""
    if not isinstance(qconfig, QConfig):
        raise TypeError(f"qconfig must be a QConfig, got {type(qconfig)}")
    if qconfig.weights is None and qconfig.input_activations is None and qconfig.output_activations is None:
        return as_model(model)
    with _on(device):
        return _quantize_sharded(model, qconfig, group, device, weight_arrays, quantize_bias, calibrate, searches)


def _quantize_sharded(model, qconfig, group, device, weight_arrays, quantize_bias, calibrate, searches):
    from .sharding import LayerSpec, quantize_sharded

    prepared = apply_pre_passes(model, qconfig, device=device, calibrate=calibrate, searches=searches, _defer_host_copies=weight_arrays is None)
    device_default = weight_arrays is None
    if device_default:
        from .seam import weight_arrays
    needs_hessian = bool(qconfig.weights.algorithm.requires_calibration)
    specs = []
    for node in prepared.targets:
        t = prepared.graph.inits[node.input[1]]
        got = prepared.meta.get(id(node), {}).get("input")
        specs.append(LayerSpec(node.input[1], int(t.dims[0]), int(t.dims[1]), tokens=int(getattr(got, "n", 0) or 0) if needs_hessian else 0,
                               hessian_key=node.input[0] if needs_hessian else ""))

    def on_this_rank(i, _spec):
        captured = []
        seam = _PackedSeam() if device_default else None        # 4-bit integers travel nibble-packed: half the gather

        def recording(value, cfg, out, nbits):
            arrays = seam(value, cfg, out, nbits) if seam is not None else weight_arrays(value, cfg, out, nbits)
            if np.asarray(value.const_value.numpy()).ndim == 2:
                packed = None if seam is None else seam.packed.pop(value.name, None)
                captured.append((packed[0],) + tuple(arrays[1:]) if packed is not None else arrays)
            return arrays

        _plan(prepared, prepared.targets[i], qconfig, recording, quantize_bias, on_device=device_default)
        (arrays,) = captured                                 # one matrix per node: its weight (a bias is a vector)
        return tuple(np.asarray(a) for a in arrays)

    gathered = quantize_sharded(specs, on_this_rank, group=group)
    if gathered is None:
        return None

    class FromTheRanks:
        """The gathered arrays as the emission's provider; integers that arrived nibble-packed go into their TensorProto as they are
        (`packed`, the protocol of `_PackedSeam`)."""

        on_device = device_default                       # (reads shapes only: a weight that lives in HBM alone stays there)

        def __init__(self):
            self.packed: dict = {}

        def __call__(self, value, cfg, out, nbits):
            w = np.asarray(value.const_value.numpy())
            if w.ndim != 2:
                return weight_arrays(value, cfg, out, nbits)       # the per-tensor bias of a QDQ Gemm: a vector, quantized here
            q, s, z = gathered[value.name]
            if device_default and not nbits and cfg.weights.dtype.bitwidth == 4 and q.ndim == 1:
                self.packed[value.name] = (q, tuple(w.shape))
                return np.broadcast_to(np.zeros((), dtype=cfg.weights.dtype.np_dtype), w.shape), s, z
            return q, s, z

    return _emit(prepared, qconfig, FromTheRanks(), quantize_bias)


def quantize_file(src, dst, qconfig: QConfig, external_data="auto", **kw) -> Message:
    """Read `src` (tensors in side files are memory-mapped, not read), quantize, write `dst`.  `external_data`: a file name
    next to `dst` for the tensors, None for one file with everything inline, "auto" (default): `<dst name>.data` when the
    source kept its tensors outside or the result would pass 1 GiB, inline otherwise.  Returns the quantized model."""
    from .onnx_proto import resolve_external_data
    src_model = load_model(src, load_external_data=False)
    had_external = resolve_external_data(src_model, os.path.dirname(os.path.abspath(os.fspath(src)))) > 0
    out = quantize_model(src_model, qconfig, **kw)
    if external_data == "auto":
        size = sum(len(t.raw_data) for t in out.graph.initializer if t.has("raw_data"))
        external_data = os.path.basename(os.fspath(dst)) + ".data" if (had_external or size > 1 << 30) else None
    save_model(out, dst, external_data=external_data)
    return out
