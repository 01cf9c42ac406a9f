"""Runs an ONNX graph on torch-ROCm so that calibration activations are born in HBM ("run the
augmented model on a ROCm / MIGraphX ORT EP (or torch-ROCm)"; reference: core/_calibration/calibrate.py:204-251).

The reference saves the augmented model to a temporary directory, opens an onnxruntime session on it and copies every
tapped activation of every batch back to the host.  `GraphRunner` walks the parsed graph (`onnx_proto`) node by node on
the GPU instead: initializers are uploaded once, each batch is one pass, the tapped values are handed to the calibration
stream as device tensors, and values nobody needs any more are dropped as the walk proceeds (a value dies after its last
consumer).  Nodes downstream of the last tapped value are not run at all -- the reference gets the same effect by
replacing the graph outputs (`_augment_model`, calibrate.py:100-124: "beneficial when we dont need the last output").

It is an interpreter for the operators transformer / MLP exports are made of, not an inference engine: an operator it
does not know raises by name.  It also runs what `quantize()` emits -- Q / DQ, DynamicQuantizeLinear, QLinearMatMul,
`com.microsoft::MatMulNBits` / `QGemm`, calls into the model's own functions -- which is how the tests execute a quantized
model end to end.  It fits `calibration_driver.run_calibration`'s runner protocol: ``runner(feed) -> {value name: tensor}``.
"""
from __future__ import annotations

from collections.abc import Mapping

import numpy as np

from .onnx_proto import _NP_OF, DataType, Message, attribute_value, tensor_to_numpy

__all__ = ["GraphRunner", "UnsupportedOperator"]


class UnsupportedOperator(NotImplementedError):
    pass


def _torch_dtype(code: int):
    import torch
    table = {DataType.FLOAT: torch.float32, DataType.UINT8: torch.uint8, DataType.INT8: torch.int8, DataType.INT16: torch.int16,
             DataType.INT32: torch.int32, DataType.INT64: torch.int64, DataType.BOOL: torch.bool, DataType.FLOAT16: torch.float16,
             DataType.DOUBLE: torch.float64, DataType.BFLOAT16: torch.bfloat16, DataType.UINT4: torch.uint8, DataType.INT4: torch.int8}
    if code not in table:
        raise UnsupportedOperator(f"GraphRunner: element type {code}")
    return table[code]


def _numpy_dtype(t):
    """NumPy's name for a tensor's element type (None for the types NumPy has no name for)."""
    import torch
    try:
        return torch.empty(0, dtype=t.dtype).numpy().dtype
    except TypeError:
        return None


def _ints(t) -> list[int]:
    """A shape-like operand as Python ints (one sync when it lives on the device; shape arithmetic is kept on the host)."""
    return [int(v) for v in (t.tolist() if t.ndim else [t.item()])]


class GraphRunner:
    """``GraphRunner(model, outputs=[names], device="cuda")(feed)`` -> ``{name: tensor on the device}``.

    model: a parsed ModelProto (`onnx_proto.parse_model`); outputs: the value names wanted (default: the graph's own
    outputs); feed: one tensor / array (single-input model) or {input name: tensor / array}."""

    takes_sink = True        # `runner(feed, sink=...)` hands every wanted value over as it is produced (calibration_driver.run_calibration)

    def __init__(self, model: Message, outputs=None, device="cuda", capture: bool = False, matmul: str = "torch", constants=None):
        """`constants`: {initializer name: tensor on the device} for initializers the caller already holds in HBM (a second
        calibration walk over the weights the first one uploaded; a weight that was rescaled on the device and whose TensorProto
        still has the old bytes): used as they are instead of the file's bytes when device, element type and shape fit."""
        import torch

        self.model, self.graph = model, model.graph
        self.device = torch.device(device)
        # capture: a calibration walk is the same few hundred small launches per batch, bound by their dispatch (~40 us per
        # node from Python).  With `capture` the second pass over a given input signature is recorded into a HIP graph and
        # every later one is a replay; a pass that cannot be recorded (a host read of device data, a host -> device copy in
        # the middle) or whose replay does not reproduce the eager pass bit for bit simply stays eager.
        self.capture = bool(capture) and self.device.type == "cuda"
        self._graphs: dict = {}              # input signature -> (graph, static inputs, static outputs) | None (stays eager)
        self._seen: set = set()
        self._moved: dict = {}               # id(host constant) -> (the constant, its copy on the device)
        self._const_nodes: dict = {}         # id(Constant node) -> its value (one object per node, so `_moved` can key on it)
        # matmul = "pieces": products of fp32 activations with a LARGE constant weight go through the library's fp16-piece GEMM
        # (`ops.matmul_pieces`: the Hessian kernels' arithmetic, 22-bit operands, fp32 accumulate: ~3 x torch's fp32 GEMM at these
        # sizes and closer to float64); the weight's pieces are made once.  Small products and everything else stay with torch.
        if matmul not in ("torch", "pieces"):
            raise ValueError(f"GraphRunner: matmul must be 'torch' or 'pieces', got {matmul!r}")
        self.matmul = matmul if self.device.type == "cuda" else "torch"
        self._weight_pieces: dict = {}       # id(constant weight) -> (the weight, its MatmulOperand)
        self.functions = {(f.domain or "", f.name, f.overload or ""): f for f in model.functions}
        self.opset = max([int(o.version or 1) for o in model.opset_import if not o.domain] or [1])
        self.wanted = list(outputs) if outputs is not None else [o.name for o in self.graph.output]
        init_names = {t.name for t in self.graph.initializer}
        self.input_names = [i.name for i in self.graph.input if i.name not in init_names]
        self.input_types = {}                                # declared element types: a feed of another type is refused, like a session does
        for i in self.graph.input:
            tt = i.type.tensor_type if i.type is not None else None
            if i.name not in init_names and tt is not None and tt.elem_type in _NP_OF:
                self.input_types[i.name] = np.dtype(_NP_OF[tt.elem_type])
        self.constants = {}
        for t in self.graph.initializer:
            held = None if constants is None else constants.get(t.name)
            if (held is not None and held.device.type == self.device.type and tuple(held.shape) == tuple(int(d) for d in t.dims)
                    and _numpy_dtype(held) == np.dtype(_NP_OF.get(t.data_type, np.void))):
                self.constants[t.name] = held
            else:
                self.constants[t.name] = self._constant(tensor_to_numpy(t), t.data_type)
        self.nodes = self._needed_nodes(self.graph.node, self.wanted)
        # the last node that reads each value: it is dropped right after
        self.last_use: dict[str, int] = {}
        for i, n in enumerate(self.nodes):
            for name in self._reads(n):
                self.last_use[name] = i

    @classmethod
    def evaluator(cls, opset: int):
        """A runner without a model: `evaluate(node, arrays)` applies ONE operator on the host (what a constant folder needs)."""
        import torch
        self = cls.__new__(cls)
        self.model = self.graph = None
        self.device = torch.device("cpu")
        self.functions, self.opset, self.wanted, self.input_names, self.constants, self.nodes, self.last_use = {}, opset, [], [], {}, [], {}
        self.capture, self._graphs, self._seen, self._moved, self._const_nodes = False, {}, set(), {}, {}
        return self

    def evaluate(self, node, arrays) -> list:
        """The outputs (NumPy) of `node` on the given input arrays (None for an absent optional input)."""
        import torch
        ins = [None if a is None else torch.from_numpy(np.array(a, order="C")) for a in arrays]
        with torch.no_grad():
            outs = self._dispatch(node, ins, {})
        return [None if o is None else (o.numpy() if isinstance(o, torch.Tensor) else np.asarray(o)) for o in outs]

    # ------------------------------------------------------------------------------------------------- preparation
    def _constant(self, a: np.ndarray, code: int | None = None):
        import torch
        if a.dtype == np.uint16 or a.dtype == np.uint32 or a.dtype == np.uint64:
            a = a.astype(np.int64)
        if a.nbytes >= 1 << 20 and a.flags.c_contiguous and a.ndim:
            import warnings
            with warnings.catch_warnings():                     # a large read-only view into the (memory-mapped) file: it is only
                warnings.filterwarnings("ignore", message="The given NumPy array is not writable")      # read, by the upload below
                return torch.from_numpy(a).to(self.device)
        t = torch.from_numpy(np.array(a, order="C"))            # a copy: 0-d stays 0-d, and views into the file are read-only
        # small integer tensors are shape arithmetic: they stay on the host, where Reshape / Slice / Expand read them
        if t.dtype == torch.int64 and t.numel() <= 64:
            return t
        return t.to(self.device)

    def _reads(self, node):
        names = [n for n in node.input if n]
        for a in node.attribute:                       # values a sub-graph captures from the enclosing scope
            if a.has("g"):
                produced = {o for n in a.g.node for o in n.output} | {i.name for i in a.g.input}
                names += [i for n in a.g.node for i in n.input if i and i not in produced]
        return names

    def _needed_nodes(self, nodes, wanted):
        producer = {}
        for n in nodes:
            for o in n.output:
                if o:
                    producer[o] = n
        need, stack = set(), [w for w in wanted]
        missing = [w for w in wanted if w not in producer and w not in self.constants and w not in self.input_names]
        if missing:
            raise KeyError(f"GraphRunner: no value named {missing[0]!r} in the graph")
        while stack:
            name = stack.pop()
            n = producer.get(name)
            if n is None or id(n) in need:
                continue
            need.add(id(n))
            stack.extend(self._reads(n))
        return [n for n in nodes if id(n) in need]       # the file's order is a topological order (ONNX requires it)

    # ------------------------------------------------------------------------------------------------- one pass
    def __call__(self, feed, sink=None) -> dict:
        """One pass.  Without `sink` the wanted values come back as a dict.  With `sink(name, tensor)` every wanted value is handed
        over the moment it exists and is not held any longer than the graph itself needs it (a 7B-width walk taps ~50 GB of
        activations per batch; consumed one by one they never coexist); the dict returned is then empty.  Recorded passes (small
        inputs) return the dictionary whatever `sink` is: their values are small, and consumers that take a whole batch of
        tensors in one grouped launch (`collect_many`, `hessian_accumulate_many`) are better served by it."""
        import torch

        if not isinstance(feed, Mapping):
            if len(self.input_names) != 1:
                raise ValueError(f"GraphRunner: the model has inputs {self.input_names}; feed them as a dict")
            feed = {self.input_names[0]: feed}
        for name in self.input_names:
            if name not in feed:
                raise KeyError(f"GraphRunner: no data for model input '{name}'")
        inputs = {}
        for name, v in feed.items():
            if name not in self.input_names:
                raise ValueError(f"GraphRunner: '{name}' is not an input of the model (inputs: {self.input_names})")
            t = v if isinstance(v, torch.Tensor) else torch.from_numpy(np.array(v, order="C"))
            declared = self.input_types.get(name)
            if declared is not None and _numpy_dtype(t) != declared:
                raise TypeError(f"GraphRunner: model input '{name}' is declared {declared}, the data handed in is {t.dtype}")
            inputs[name] = t.to(self.device, non_blocking=True)
        if not self.capture or sum(t.numel() for t in inputs.values()) > _CAPTURE_MAX_INPUT_ELEMENTS:
            return self._eager(inputs, sink)                  # (a large pass is bound by its kernels, not by their dispatch)
        return self._recorded(inputs)                         # small passes: the whole dictionary at once (one grouped launch for its consumers)

    def _recorded(self, inputs) -> dict:
        key = tuple(sorted((name, tuple(t.shape), str(t.dtype)) for name, t in inputs.items()))
        entry = self._graphs.get(key, False)
        if entry:
            graph, static_in, static_out = entry
            for name, t in inputs.items():
                static_in[name].copy_(t)
            graph.replay()
            return {name: t.clone() for name, t in static_out.items()}      # the static buffers are overwritten by the next replay
        if entry is None or key not in self._seen:                           # first sight of a signature (or not recordable): eager
            self._seen.add(key)
            return self._eager(inputs)
        return self._record(key, inputs)

    def _eager(self, inputs, sink=None) -> dict:
        import torch
        env = dict(self.constants)
        env.update(inputs)
        if sink is None:
            with torch.no_grad():
                self._run(self.nodes, env, set(self.wanted), self.last_use)
            return {name: env[name] for name in self.wanted}
        pending = set(self.wanted)
        for name in [w for w in self.wanted if w in env]:      # model inputs / constants that are themselves wanted
            sink(name, env[name])
            pending.discard(name)
        with torch.no_grad():
            self._run(self.nodes, env, (), self.last_use, sink=sink, pending=pending)
        return {}

    def _record(self, key, inputs) -> dict:
        """Record one pass into a HIP graph, check one replay of it against the eager pass on the same inputs, keep it only
        if every tapped value is bit-identical.  Returns this batch's values either way."""
        import torch
        eager = self._eager(inputs)
        self._graphs[key] = None
        if not all(isinstance(t, torch.Tensor) and t.is_cuda for t in eager.values()):
            return eager
        if all(name in inputs or name in self.constants for name in self.wanted):
            return eager                                                   # nothing is computed for these: nothing to record
        static_in = {name: t.clone() for name, t in inputs.items()}
        graph = torch.cuda.CUDAGraph()
        try:
            torch.cuda.synchronize()
            import warnings
            with warnings.catch_warnings():
                warnings.filterwarnings("ignore", message="The CUDA Graph is empty")      # tapped values that are views of the feed
                with torch.cuda.graph(graph):
                    static_out = self._eager(static_in)
            graph.replay()
            torch.cuda.synchronize()
            same = all(torch.equal(static_out[name], eager[name]) for name in eager)
        except Exception as exc:                                           # noqa: BLE001 -- whatever refused the recording: stay eager
            import logging
            logging.getLogger("onnx_quantize").debug("GraphRunner: pass not recordable (%s); staying eager", str(exc).splitlines()[0][:120])
            torch.cuda.synchronize()
            return eager
        if same:
            self._graphs[key] = (graph, static_in, static_out)
        return eager

    def _run(self, nodes, env, keep=(), last_use=None, sink=None, pending=()):
        for i, node in enumerate(nodes):
            ins = [env[n] if n else None for n in node.input]
            outs = self._dispatch(node, ins, env)
            for name, value in zip(node.output, outs):
                if name:
                    env[name] = value
                    if name in pending:                       # consumed now; it lives on only as long as later nodes read it
                        sink(name, value)
                        if last_use is not None and name not in last_use:
                            del env[name]
            if last_use is not None:
                for name in node.input:
                    if name and last_use.get(name) == i and name not in keep and name not in self.constants:
                        env.pop(name, None)

    def _dispatch(self, node, ins, env):
        import torch

        domain = node.domain or ""
        fn = self.functions.get((domain, node.op_type, node.overload or ""))
        if fn is not None:
            return self._call_function(fn, node, ins)
        if domain not in ("", "ai.onnx", "com.microsoft"):
            raise UnsupportedOperator(f"GraphRunner: operator {domain}::{node.op_type} (node '{node.name}') and no function of that name in the model")
        impl = getattr(self, "_op_" + node.op_type, None)
        if impl is None:
            raise UnsupportedOperator(f"GraphRunner: operator '{node.op_type}' (node '{node.name}') is not implemented")
        # shape arithmetic lives on the host; an operator that mixes it with device data takes it to the device
        devs = {t.device.type for t in ins if isinstance(t, torch.Tensor)}
        if len(devs) > 1 and node.op_type not in _HOST_OPERANDS:
            ins = [self._to_device(t, node.op_type) if isinstance(t, torch.Tensor) else t for t in ins]
        attrs = {a.name: a for a in node.attribute}
        out = impl(node, ins, _Attrs(attrs), env)
        return out if isinstance(out, (tuple, list)) else (out,)

    def _to_device(self, t, op_type):
        """A host operand of an operator that also has device operands.  One-element shape arithmetic joins elementwise
        arithmetic as a Python number (no copy at all: the value is a property of the input shapes); a constant is copied
        once and the copy reused; anything else is copied now."""
        if t.device.type == self.device.type:
            return t
        if t.numel() == 1 and op_type in _SCALAR_FRIENDLY:
            return t.item()
        hit = self._moved.get(id(t))
        if hit is not None and hit[0] is t:
            return hit[1]
        moved = t.to(self.device)
        if any(t is c for c in self.constants.values()) or any(t is c for c in self._const_nodes.values()):
            self._moved[id(t)] = (t, moved)
        return moved

    def _call_function(self, fn, node, ins):
        if fn.attribute or fn.attribute_proto:
            given = {a.name for a in node.attribute}
            if any(r.ref_attr_name for n in fn.node for r in n.attribute):
                raise UnsupportedOperator(f"GraphRunner: function '{fn.name}' refers to attributes {sorted(given)} of its call")
        env = {}
        for name, value in zip(fn.input, ins):
            env[name] = value
        for name in list(fn.input)[len(ins):]:
            env[name] = None
        self._run(list(fn.node), env)          # a function body sees its inputs only (ONNX functions are closed)
        return tuple(env[o] for o in fn.output)

    # ------------------------------------------------------------------------------------------------- operators
    # elementwise
    def _op_Add(self, n, x, a, e): return x[0] + x[1]
    def _op_Sub(self, n, x, a, e): return x[0] - x[1]
    def _op_Mul(self, n, x, a, e): return x[0] * x[1]

    def _op_Div(self, n, x, a, e):
        import torch
        floating = [t.is_floating_point() if isinstance(t, torch.Tensor) else isinstance(t, float) for t in x[:2]]
        if not any(floating):                                 # integer division truncates towards zero
            return torch.div(x[0], x[1], rounding_mode="trunc") if isinstance(x[0], torch.Tensor) else \
                torch.div(torch.tensor(x[0], device=x[1].device), x[1], rounding_mode="trunc")
        return x[0] / x[1]

    def _op_Pow(self, n, x, a, e):
        import torch
        base, exp = x[0], x[1]
        if not isinstance(base, torch.Tensor):                 # a host scalar raised to a device tensor
            base = torch.full_like(exp, base, dtype=exp.dtype if exp.is_floating_point() else torch.float32)
        if isinstance(exp, torch.Tensor) and base.is_floating_point():
            exp = exp.to(base.dtype)
        return torch.pow(base, exp)

    def _op_Sqrt(self, n, x, a, e): return x[0].sqrt()
    def _op_Exp(self, n, x, a, e): return x[0].exp()
    def _op_Log(self, n, x, a, e): return x[0].log()
    def _op_Neg(self, n, x, a, e): return -x[0]
    def _op_Abs(self, n, x, a, e): return x[0].abs()
    def _op_Erf(self, n, x, a, e): return x[0].erf()
    def _op_Tanh(self, n, x, a, e): return x[0].tanh()
    def _op_Sin(self, n, x, a, e): return x[0].sin()
    def _op_Cos(self, n, x, a, e): return x[0].cos()
    def _op_Sigmoid(self, n, x, a, e): return x[0].sigmoid()
    def _op_Relu(self, n, x, a, e): return x[0].relu()
    def _op_Reciprocal(self, n, x, a, e): return x[0].reciprocal()
    def _op_Floor(self, n, x, a, e): return x[0].floor()
    def _op_Ceil(self, n, x, a, e): return x[0].ceil()
    def _op_Round(self, n, x, a, e): return x[0].round()
    def _op_Not(self, n, x, a, e): return ~x[0]
    def _op_And(self, n, x, a, e): return x[0] & x[1]
    def _op_Or(self, n, x, a, e): return x[0] | x[1]
    def _op_Equal(self, n, x, a, e): return x[0] == x[1]
    def _op_Less(self, n, x, a, e): return x[0] < x[1]
    def _op_Greater(self, n, x, a, e): return x[0] > x[1]
    def _op_LessOrEqual(self, n, x, a, e): return x[0] <= x[1]
    def _op_GreaterOrEqual(self, n, x, a, e): return x[0] >= x[1]
    def _op_Identity(self, n, x, a, e): return x[0]
    def _op_Dropout(self, n, x, a, e): return (x[0], None)[:max(1, len(n.output))]      # inference: identity (+ an unused mask)

    def _op_Where(self, n, x, a, e):
        import torch
        return torch.where(x[0], x[1], x[2])

    def _op_Min(self, n, x, a, e):
        import torch
        out = x[0]
        for t in x[1:]:
            out = torch.minimum(out, t)
        return out

    def _op_Max(self, n, x, a, e):
        import torch
        out = x[0]
        for t in x[1:]:
            out = torch.maximum(out, t)
        return out

    def _op_Sum(self, n, x, a, e):
        out = x[0]
        for t in x[1:]:
            out = out + t
        return out

    def _op_LeakyRelu(self, n, x, a, e):
        import torch
        return torch.nn.functional.leaky_relu(x[0], a.get("alpha", 0.01))

    def _op_Gelu(self, n, x, a, e):
        import torch
        return torch.nn.functional.gelu(x[0], approximate=a.get("approximate", "none"))

    def _op_Softplus(self, n, x, a, e):
        import torch
        return torch.nn.functional.softplus(x[0])

    def _op_Clip(self, n, x, a, e):
        lo = x[1] if len(x) > 1 and x[1] is not None else a.get("min", None)
        hi = x[2] if len(x) > 2 and x[2] is not None else a.get("max", None)
        out = x[0]
        if lo is not None:
            out = out.clamp(min=lo.item() if hasattr(lo, "item") else lo)
        if hi is not None:
            out = out.clamp(max=hi.item() if hasattr(hi, "item") else hi)
        return out

    def _op_Cast(self, n, x, a, e):
        return x[0].to(_torch_dtype(a["to"]))

    def _op_Softmax(self, n, x, a, e):
        import torch
        if self.opset >= 13:
            return torch.softmax(x[0], dim=a.get("axis", -1))
        axis = a.get("axis", 1)                              # before opset 13: over everything from `axis` on, flattened
        axis = axis if axis >= 0 else axis + x[0].ndim
        return torch.softmax(x[0].flatten(axis), dim=-1).reshape(x[0].shape)

    def _op_LogSoftmax(self, n, x, a, e):
        import torch
        return torch.log_softmax(x[0], dim=a.get("axis", -1))

    def _op_LayerNormalization(self, n, x, a, e):
        import torch
        axis = a.get("axis", -1)
        t = x[0]
        axis = axis if axis >= 0 else axis + t.ndim
        out = torch.nn.functional.layer_norm(t, tuple(t.shape[axis:]), x[1], x[2] if len(x) > 2 else None, a.get("epsilon", 1e-5))
        if len([o for o in n.output if o]) > 1:
            raise UnsupportedOperator("GraphRunner: LayerNormalization with Mean / InvStdDev outputs")
        return out

    def _reduce(self, fn, n, x, a):
        axes = _ints(x[1]) if len(x) > 1 and x[1] is not None else a.get("axes", None)
        keep = bool(a.get("keepdims", 1))
        if not axes:
            if a.get("noop_with_empty_axes", 0) and len(x) > 1:
                return x[0]
            axes = list(range(x[0].ndim))
        return fn(x[0], axes, keep)

    def _op_ReduceMean(self, n, x, a, e): return self._reduce(lambda t, ax, k: t.mean(dim=ax, keepdim=k), n, x, a)
    def _op_ReduceSum(self, n, x, a, e): return self._reduce(lambda t, ax, k: t.sum(dim=ax, keepdim=k), n, x, a)
    def _op_ReduceMax(self, n, x, a, e): return self._reduce(lambda t, ax, k: t.amax(dim=ax, keepdim=k), n, x, a)
    def _op_ReduceMin(self, n, x, a, e): return self._reduce(lambda t, ax, k: t.amin(dim=ax, keepdim=k), n, x, a)

    def _op_ReduceProd(self, n, x, a, e):
        def prod(t, axes, keep):
            for ax in sorted((ax % t.ndim for ax in axes), reverse=True):
                t = t.prod(dim=ax, keepdim=keep)
            return t
        return self._reduce(prod, n, x, a)

    def _op_ReduceL1(self, n, x, a, e): return self._reduce(lambda t, ax, k: t.abs().sum(dim=ax, keepdim=k), n, x, a)
    def _op_ReduceL2(self, n, x, a, e): return self._reduce(lambda t, ax, k: (t * t).sum(dim=ax, keepdim=k).sqrt(), n, x, a)
    def _op_ReduceSumSquare(self, n, x, a, e): return self._reduce(lambda t, ax, k: (t * t).sum(dim=ax, keepdim=k), n, x, a)
    def _op_ReduceLogSum(self, n, x, a, e): return self._reduce(lambda t, ax, k: t.sum(dim=ax, keepdim=k).log(), n, x, a)
    def _op_ReduceLogSumExp(self, n, x, a, e): return self._reduce(lambda t, ax, k: t.logsumexp(dim=ax, keepdim=k), n, x, a)

    def _op_ArgMin(self, n, x, a, e):
        if a.get("select_last_index", 0):
            raise UnsupportedOperator("GraphRunner: ArgMin with select_last_index")
        return x[0].argmin(dim=a.get("axis", 0), keepdim=bool(a.get("keepdims", 1)))

    def _op_TopK(self, n, x, a, e):
        import torch
        k = _ints(x[1])[0] if len(x) > 1 else a["k"]
        values, indices = torch.topk(x[0], k, dim=a.get("axis", -1), largest=bool(a.get("largest", 1)), sorted=True)
        return values, indices

    def _op_NonZero(self, n, x, a, e):
        return x[0].nonzero().t().contiguous()              # (a data-dependent shape: such a pass is never recorded)

    def _op_OneHot(self, n, x, a, e):
        import torch
        depth = _ints(x[1])[0]
        axis = a.get("axis", -1)
        idx = x[0].to(torch.int64)
        idx = torch.where(idx < 0, idx + depth, idx)
        hot = torch.nn.functional.one_hot(idx.clamp(0, depth - 1), depth).to(torch.bool) & ((idx >= 0) & (idx < depth)).unsqueeze(-1)
        values = x[2].to(x[0].device) if x[2].device != x[0].device else x[2]
        out = torch.where(hot, values[1], values[0])
        if axis not in (-1, idx.ndim):
            out = out.movedim(-1, axis if axis >= 0 else axis + idx.ndim + 1)
        return out

    def _op_ScatterElements(self, n, x, a, e):
        import torch
        t, idx, upd = x[0], x[1].to(torch.int64), x[2]
        axis = a.get("axis", 0)
        idx = torch.where(idx < 0, idx + t.shape[axis], idx)
        reduction = a.get("reduction", "none")
        if reduction == "none":
            return t.scatter(axis, idx, upd)
        how = {"add": "sum", "mul": "prod", "max": "amax", "min": "amin"}.get(reduction)
        if how is None:
            raise UnsupportedOperator(f"GraphRunner: ScatterElements with reduction '{reduction}'")
        return t.scatter_reduce(axis, idx, upd, how, include_self=True)

    def _op_ScatterND(self, n, x, a, e):
        import torch
        t, idx, upd = x[0].clone(), x[1].to(torch.int64), x[2]
        reduction = a.get("reduction", "none")
        if reduction not in ("none", "add"):
            raise UnsupportedOperator(f"GraphRunner: ScatterND with reduction '{reduction}'")
        last = idx.shape[-1]
        flat = idx.reshape(-1, last)
        flat = torch.where(flat < 0, flat + torch.tensor(t.shape[:last], device=flat.device), flat)
        t.index_put_(tuple(flat[:, j] for j in range(last)), upd.reshape(flat.shape[0], *t.shape[last:]), accumulate=reduction == "add")
        return t

    def _op_GatherND(self, n, x, a, e):
        import torch
        t, idx = x[0], x[1].to(torch.int64)
        b = a.get("batch_dims", 0)
        if b:
            raise UnsupportedOperator("GraphRunner: GatherND with batch_dims")
        last = idx.shape[-1]
        flat = idx.reshape(-1, last)
        flat = torch.where(flat < 0, flat + torch.tensor(t.shape[:last], device=flat.device), flat)
        return t[tuple(flat[:, j] for j in range(last))].reshape(*idx.shape[:-1], *t.shape[last:])

    def _op_DepthToSpace(self, n, x, a, e):
        t, bs = x[0], a["blocksize"]
        b, c, h, w = t.shape
        if a.get("mode", "DCR") == "DCR":
            t = t.reshape(b, bs, bs, c // (bs * bs), h, w).permute(0, 3, 4, 1, 5, 2)
        else:
            t = t.reshape(b, c // (bs * bs), bs, bs, h, w).permute(0, 1, 4, 2, 5, 3)
        return t.reshape(b, c // (bs * bs), h * bs, w * bs)

    def _op_SpaceToDepth(self, n, x, a, e):
        t, bs = x[0], a["blocksize"]
        b, c, h, w = t.shape
        return t.reshape(b, c, h // bs, bs, w // bs, bs).permute(0, 3, 5, 1, 2, 4).reshape(b, c * bs * bs, h // bs, w // bs)

    def _op_InstanceNormalization(self, n, x, a, e):
        import torch
        return torch.nn.functional.instance_norm(x[0], weight=x[1], bias=x[2], eps=a.get("epsilon", 1e-5))

    def _op_GroupNormalization(self, n, x, a, e):
        import torch
        if self.opset < 21:                                  # one scale / bias per group before opset 21
            raise UnsupportedOperator("GraphRunner: GroupNormalization before opset 21")
        return torch.nn.functional.group_norm(x[0], a["num_groups"], x[1], x[2], a.get("epsilon", 1e-5))

    def _op_LpNormalization(self, n, x, a, e):
        import torch
        return torch.nn.functional.normalize(x[0], p=float(a.get("p", 2)), dim=a.get("axis", -1), eps=0.0)

    def _op_ConvTranspose(self, n, x, a, e):
        import torch
        t, w = x[0], x[1]
        bias = x[2] if len(x) > 2 else None
        spatial = t.ndim - 2
        strides, dilations = a.get("strides", [1] * spatial), a.get("dilations", [1] * spatial)
        if a.get("auto_pad", "NOTSET") != "NOTSET" or a.get("output_shape", None) is not None:
            raise UnsupportedOperator("GraphRunner: ConvTranspose with auto_pad / output_shape")
        pads = a.get("pads", [0] * (2 * spatial))
        begin, end = list(pads[:spatial]), list(pads[spatial:])
        out_pad = list(a.get("output_padding", [0] * spatial))
        fn = {1: torch.nn.functional.conv_transpose1d, 2: torch.nn.functional.conv_transpose2d, 3: torch.nn.functional.conv_transpose3d}[spatial]
        if begin == end:
            return fn(t, w, bias, stride=strides, padding=begin, output_padding=out_pad, groups=a.get("group", 1), dilation=dilations)
        out = fn(t, w, bias, stride=strides, padding=0, output_padding=out_pad, groups=a.get("group", 1), dilation=dilations)
        index = [slice(None)] * 2 + [slice(b, out.shape[2 + i] - en) for i, (b, en) in enumerate(zip(begin, end))]
        return out[tuple(index)]

    # linear algebra
    def _op_MatMul(self, n, x, a, e):
        import torch
        w = x[1]
        if (self.matmul == "pieces" and w.ndim == 2 and w.dtype == torch.float32 and x[0].dtype == torch.float32 and x[0].is_cuda
                and min(w.shape) >= _PIECES_MIN_WIDTH and x[0].numel() // max(1, w.shape[0]) >= _PIECES_MIN_ROWS and x[0].ndim >= 2):
            hit = self._weight_pieces.get(id(w))
            if hit is None and any(w is c for c in self.constants.values()):
                from .hip import ops
                # the pieces are a second copy of the weight (two fp16 halves): kept only while HBM has room for them and the
                # activations of a pass; past that the product stays with torch's fp32 GEMM on the weight itself
                free = torch.cuda.mem_get_info(w.device)[0] + torch.cuda.memory_reserved(w.device) - torch.cuda.memory_allocated(w.device)
                hit = self._weight_pieces[id(w)] = (w, ops.matmul_prepare(w, False) if free > _PIECES_HEADROOM * w.numel() * 4 else None)
            if hit is not None and hit[0] is w and hit[1] is not None:
                from .hip import ops
                return ops.matmul_pieces(x[0], hit[1])
        return torch.matmul(x[0], w)

    def _op_FusedMatMul(self, n, x, a, e):                  # com.microsoft: alpha * op(A) @ op(B), the transposes on the last two axes
        import torch
        if a.get("transBatchA", 0) or a.get("transBatchB", 0):
            raise UnsupportedOperator("GraphRunner: FusedMatMul with transBatchA / transBatchB")
        A = x[0].transpose(-1, -2) if a.get("transA", 0) and x[0].ndim > 1 else x[0]
        B = x[1].transpose(-1, -2) if a.get("transB", 0) and x[1].ndim > 1 else x[1]
        out = torch.matmul(A, B)
        return out * a.get("alpha", 1.0) if a.get("alpha", 1.0) != 1.0 else out

    def _op_Gemm(self, n, x, a, e):
        A = x[0].t() if a.get("transA", 0) else x[0]
        B = x[1].t() if a.get("transB", 0) else x[1]
        out = (A @ B) * a.get("alpha", 1.0) if a.get("alpha", 1.0) != 1.0 else A @ B
        if len(x) > 2 and x[2] is not None:
            out = out + (x[2] * a.get("beta", 1.0) if a.get("beta", 1.0) != 1.0 else x[2])
        return out

    # convolutional front ends of models whose classifier is a Gemm / MatMul (the calibration walk has to get through them)
    @staticmethod
    def _pads(a, spatial, x_shape=None, kernel=None, strides=None, dilations=None):
        pads = a.get("pads", None)
        auto = a.get("auto_pad", "NOTSET")
        if auto in ("SAME_UPPER", "SAME_LOWER"):
            total = []
            for i in range(spatial):
                size, k, st, d = x_shape[2 + i], kernel[i], strides[i], dilations[i]
                out = -(-size // st)
                total.append(max(0, (out - 1) * st + (k - 1) * d + 1 - size))
            begin = [t // 2 if auto == "SAME_UPPER" else t - t // 2 for t in total]
            return begin, [t - b for t, b in zip(total, begin)]
        if pads is None or auto == "VALID":
            return [0] * spatial, [0] * spatial
        return list(pads[:spatial]), list(pads[spatial:])

    def _op_Conv(self, n, x, a, e):
        import torch
        t, w = x[0], x[1]
        bias = x[2] if len(x) > 2 else None
        spatial = t.ndim - 2
        strides, dilations = a.get("strides", [1] * spatial), a.get("dilations", [1] * spatial)
        begin, end = self._pads(a, spatial, t.shape, list(w.shape[2:]), strides, dilations)
        if begin != end:
            pad = []
            for b, en in zip(reversed(begin), reversed(end)):
                pad += [b, en]
            t, begin = torch.nn.functional.pad(t, pad), [0] * spatial
        fn = {1: torch.nn.functional.conv1d, 2: torch.nn.functional.conv2d, 3: torch.nn.functional.conv3d}[spatial]
        return fn(t, w, bias, stride=strides, padding=begin, dilation=dilations, groups=a.get("group", 1))

    def _pool(self, n, x, a, kind):
        import torch
        t = x[0]
        spatial = t.ndim - 2
        kernel = a["kernel_shape"]
        strides, dilations = a.get("strides", [1] * spatial), a.get("dilations", [1] * spatial)
        begin, end = self._pads(a, spatial, t.shape, kernel, strides, dilations)
        ceil = bool(a.get("ceil_mode", 0))
        if begin != end:
            raise UnsupportedOperator(f"GraphRunner: {kind} with asymmetric padding")
        if kind == "MaxPool":
            fn = {1: torch.nn.functional.max_pool1d, 2: torch.nn.functional.max_pool2d, 3: torch.nn.functional.max_pool3d}[spatial]
            return fn(t, kernel, strides, begin, dilations, ceil_mode=ceil)
        fn = {1: torch.nn.functional.avg_pool1d, 2: torch.nn.functional.avg_pool2d, 3: torch.nn.functional.avg_pool3d}[spatial]
        return fn(t, kernel, strides, begin, ceil_mode=ceil, count_include_pad=bool(a.get("count_include_pad", 0)))

    def _op_MaxPool(self, n, x, a, e):
        if len([o for o in n.output if o]) > 1:
            raise UnsupportedOperator("GraphRunner: MaxPool with the Indices output")
        return self._pool(n, x, a, "MaxPool")

    def _op_AveragePool(self, n, x, a, e): return self._pool(n, x, a, "AveragePool")
    def _op_GlobalAveragePool(self, n, x, a, e): return x[0].mean(dim=tuple(range(2, x[0].ndim)), keepdim=True)
    def _op_GlobalMaxPool(self, n, x, a, e): return x[0].amax(dim=tuple(range(2, x[0].ndim)), keepdim=True)

    def _op_BatchNormalization(self, n, x, a, e):
        import torch
        if a.get("training_mode", 0):
            raise UnsupportedOperator("GraphRunner: BatchNormalization in training mode")
        return torch.nn.functional.batch_norm(x[0], x[3], x[4], x[1], x[2], False, 0.0, a.get("epsilon", 1e-5))

    def _op_Pad(self, n, x, a, e):
        import torch
        t = x[0]
        pads = a["pads"] if len(x) < 2 else _ints(x[1])                          # opset < 11: attributes
        value = a.get("value", 0.0) if len(x) < 2 else (x[2].item() if len(x) > 2 and x[2] is not None and x[2].numel() else 0)
        axes = _ints(x[3]) if len(x) > 3 and x[3] is not None else list(range(t.ndim))
        mode = a.get("mode", "constant")
        begin, end = [0] * t.ndim, [0] * t.ndim
        for j, ax in enumerate(axes):
            begin[ax], end[ax] = pads[j], pads[len(axes) + j]
        for ax in range(t.ndim):                                                 # negative pads crop
            lo, hi = max(0, -begin[ax]), t.shape[ax] - max(0, -end[ax])
            if lo or hi != t.shape[ax]:
                t = t.narrow(ax, lo, max(0, hi - lo))
        begin, end = [max(0, b) for b in begin], [max(0, b) for b in end]
        if not any(begin) and not any(end):
            return t
        flat = [v for ax in reversed(range(t.ndim)) for v in (begin[ax], end[ax])]   # torch: last axis first
        if mode == "constant":
            return torch.nn.functional.pad(t, flat, value=value)
        torch_mode = {"reflect": "reflect", "edge": "replicate", "wrap": "circular"}.get(mode)
        padded = [ax for ax in range(t.ndim) if begin[ax] or end[ax]]
        if torch_mode is None or not padded or padded[0] < t.ndim - 3 or t.ndim < 2:
            raise UnsupportedOperator(f"GraphRunner: Pad (node '{n.name}') with mode '{mode}' on axes {padded} of a rank-{t.ndim} value")
        spatial = t.ndim - padded[0]                                             # torch pads the last 1-3 axes of a batched value
        lead = t.shape[:t.ndim - spatial]
        out = torch.nn.functional.pad(t.reshape(1, -1, *t.shape[t.ndim - spatial:]), flat[:2 * spatial], mode=torch_mode)
        return out.reshape(*lead, *out.shape[2:])

    def _op_Resize(self, n, x, a, e):
        import torch
        t = x[0]
        scales = x[2] if len(x) > 2 and x[2] is not None and x[2].numel() else None
        sizes = _ints(x[3]) if len(x) > 3 and x[3] is not None and x[3].numel() else None
        if a.get("axes", None) is not None or a.get("antialias", 0) or a.get("keep_aspect_ratio_policy", "stretch") != "stretch":
            raise UnsupportedOperator(f"GraphRunner: Resize (node '{n.name}') with axes / antialias / keep_aspect_ratio_policy")
        if sizes is None:
            if scales is None:
                raise ValueError(f"node '{n.name}': Resize needs scales or sizes")
            factors = [float(v) for v in scales.tolist()]
            sizes = [int(d * f) for d, f in zip(t.shape, factors)]              # floor, like the operator
        else:
            factors = None
        if t.ndim < 3 or list(sizes[:2]) != list(t.shape[:2]):
            raise UnsupportedOperator(f"GraphRunner: Resize (node '{n.name}') of the batch or channel axis")
        mode, coord = a.get("mode", "nearest"), a.get("coordinate_transformation_mode", "half_pixel")
        out_size = sizes[2:]
        if mode == "nearest":
            if coord == "asymmetric" and a.get("nearest_mode", "round_prefer_floor") == "floor":
                # torch's "nearest" is floor(i * scale); it must be handed the operator's scale, not the ratio of the sizes
                kw = dict(scale_factor=factors[2:], recompute_scale_factor=False) if factors else dict(size=out_size)
                return torch.nn.functional.interpolate(t, mode="nearest", **kw)
            raise UnsupportedOperator(f"GraphRunner: Resize (node '{n.name}') nearest with {coord} / {a.get('nearest_mode', 'round_prefer_floor')}")
        if mode in ("linear", "cubic"):
            torch_mode = {("linear", 1): "linear", ("linear", 2): "bilinear", ("linear", 3): "trilinear", ("cubic", 2): "bicubic"}.get((mode, t.ndim - 2))
            if torch_mode is None or coord not in ("half_pixel", "pytorch_half_pixel", "align_corners") or \
                    (mode == "cubic" and abs(a.get("cubic_coeff_a", -0.75) + 0.75) > 1e-6):
                raise UnsupportedOperator(f"GraphRunner: Resize (node '{n.name}') {mode} with {coord} on {t.ndim - 2} axes")
            if coord == "pytorch_half_pixel" and any(v == 1 for v in out_size):
                raise UnsupportedOperator(f"GraphRunner: Resize (node '{n.name}') pytorch_half_pixel to a length of one")
            kw = dict(scale_factor=factors[2:], recompute_scale_factor=False) if factors and coord != "align_corners" else dict(size=out_size)
            return torch.nn.functional.interpolate(t, mode=torch_mode, align_corners=coord == "align_corners", **kw)
        raise UnsupportedOperator(f"GraphRunner: Resize (node '{n.name}') with mode '{mode}'")

    def _op_Selu(self, n, x, a, e):
        import torch
        alpha, gamma = a.get("alpha", 1.6732632423543772), a.get("gamma", 1.0507009873554805)
        return gamma * torch.where(x[0] > 0, x[0], alpha * (x[0].exp() - 1))

    def _op_Celu(self, n, x, a, e):
        import torch
        return torch.nn.functional.celu(x[0], a.get("alpha", 1.0))

    def _op_Mish(self, n, x, a, e):
        import torch
        return torch.nn.functional.mish(x[0])

    def _op_Softsign(self, n, x, a, e): return x[0] / (1 + x[0].abs())

    def _op_ThresholdedRelu(self, n, x, a, e):
        import torch
        return torch.where(x[0] > a.get("alpha", 1.0), x[0], torch.zeros_like(x[0]))

    def _op_Shrink(self, n, x, a, e):
        import torch
        lambd, bias = a.get("lambd", 0.5), a.get("bias", 0.0)
        return torch.where(x[0] < -lambd, x[0] + bias, torch.where(x[0] > lambd, x[0] - bias, torch.zeros_like(x[0])))

    def _op_Xor(self, n, x, a, e): return x[0] ^ x[1]
    def _op_Tan(self, n, x, a, e): return x[0].tan()
    def _op_Atan(self, n, x, a, e): return x[0].atan()
    def _op_Asin(self, n, x, a, e): return x[0].asin()
    def _op_Acos(self, n, x, a, e): return x[0].acos()
    def _op_Sinh(self, n, x, a, e): return x[0].sinh()
    def _op_Cosh(self, n, x, a, e): return x[0].cosh()
    def _op_Mean(self, n, x, a, e): return sum(x[1:], x[0]) / len(x)

    def _op_HardSwish(self, n, x, a, e):
        import torch
        return torch.nn.functional.hardswish(x[0])

    def _op_PRelu(self, n, x, a, e):
        import torch
        return torch.where(x[0] >= 0, x[0], x[0] * x[1])

    def _op_Einsum(self, n, x, a, e):
        import torch
        return torch.einsum(a["equation"], *x)

    def _op_CumSum(self, n, x, a, e):
        import torch
        axis = int(x[1].item())
        t = x[0].flip(axis) if a.get("reverse", 0) else x[0]
        out = torch.cumsum(t, dim=axis)
        if a.get("exclusive", 0):
            out = out - t
        return out.flip(axis) if a.get("reverse", 0) else out

    def _op_ArgMax(self, n, x, a, e):
        import torch
        return torch.argmax(x[0], dim=a.get("axis", 0), keepdim=bool(a.get("keepdims", 1)))

    def _op_GatherElements(self, n, x, a, e):
        import torch
        axis = a.get("axis", 0)
        idx = x[1].to(torch.int64)
        return torch.gather(x[0], axis, torch.where(idx < 0, idx + x[0].shape[axis], idx))

    def _op_IsNaN(self, n, x, a, e): return x[0].isnan()
    def _op_IsInf(self, n, x, a, e): return x[0].isinf()
    def _op_Sign(self, n, x, a, e): return x[0].sign()
    def _op_Mod(self, n, x, a, e):
        import torch
        return torch.fmod(x[0], x[1]) if a.get("fmod", 0) else torch.remainder(x[0], x[1])

    def _op_HardSigmoid(self, n, x, a, e): return (x[0] * a.get("alpha", 0.2) + a.get("beta", 0.5)).clamp(0, 1)

    def _op_Elu(self, n, x, a, e):
        import torch
        return torch.nn.functional.elu(x[0], a.get("alpha", 1.0))

    # shapes
    def _op_Shape(self, n, x, a, e):
        import torch
        dims = list(x[0].shape)
        start, end = a.get("start", 0), a.get("end", None)
        return torch.tensor(dims[start:end], dtype=torch.int64)

    def _op_Size(self, n, x, a, e):
        import torch
        return torch.tensor(x[0].numel(), dtype=torch.int64)

    def _op_Constant(self, n, x, a, e):
        hit = self._const_nodes.get(id(n))
        if hit is None:
            hit = self._const_nodes[id(n)] = self._constant_value(n, a)
        return hit

    def _constant_value(self, n, a):
        import torch
        if "value" in a:
            return self._constant(a["value"])
        if "value_int" in a:
            return torch.tensor(a["value_int"], dtype=torch.int64)
        if "value_ints" in a:
            return torch.tensor(a["value_ints"], dtype=torch.int64)
        if "value_float" in a:
            return torch.tensor(a["value_float"], dtype=torch.float32, device=self.device)
        if "value_floats" in a:
            return torch.tensor(a["value_floats"], dtype=torch.float32, device=self.device)
        raise UnsupportedOperator(f"GraphRunner: Constant node '{n.name}' with none of the value attributes this runner reads")

    def _op_ConstantOfShape(self, n, x, a, e):
        import torch
        value = a.get("value", None)
        shape = _ints(x[0])
        if value is None:
            return torch.zeros(shape, dtype=torch.float32, device=self.device)
        v = torch.from_numpy(np.array(value, order="C")).reshape(-1)[0]
        out = torch.full(shape, v.item(), dtype=v.dtype)
        return out if (out.dtype == torch.int64 and out.numel() <= 64) else out.to(self.device)

    def _op_Reshape(self, n, x, a, e):
        shape = _ints(x[1])
        if not a.get("allowzero", 0):
            shape = [x[0].shape[i] if d == 0 else d for i, d in enumerate(shape)]
        return x[0].reshape(shape)

    def _op_Flatten(self, n, x, a, e):
        axis = a.get("axis", 1)
        axis = axis if axis >= 0 else axis + x[0].ndim
        return x[0].reshape(int(np.prod(x[0].shape[:axis], dtype=np.int64)), -1)

    def _op_Transpose(self, n, x, a, e):
        perm = a.get("perm", None)
        return x[0].permute(perm if perm is not None else list(range(x[0].ndim))[::-1])

    def _axes(self, x, a):
        return _ints(x[1]) if len(x) > 1 and x[1] is not None else a.get("axes", None)

    def _op_Unsqueeze(self, n, x, a, e):
        axes = self._axes(x, a)
        rank = x[0].ndim + len(axes)
        out = x[0]
        for ax in sorted(ax if ax >= 0 else ax + rank for ax in axes):
            out = out.unsqueeze(ax)
        return out

    def _op_Squeeze(self, n, x, a, e):
        axes = self._axes(x, a)
        if axes is None:
            return x[0].squeeze()
        out = x[0]
        for ax in sorted((ax if ax >= 0 else ax + x[0].ndim for ax in axes), reverse=True):
            out = out.squeeze(ax)
        return out

    def _op_Concat(self, n, x, a, e):
        import torch
        return torch.cat([t for t in x if t is not None], dim=a["axis"])

    def _op_Split(self, n, x, a, e):
        import torch
        axis = a.get("axis", 0)
        sizes = _ints(x[1]) if len(x) > 1 and x[1] is not None else a.get("split", None)
        if sizes is None:
            parts = a.get("num_outputs", None) or len(n.output)
            size = -(-x[0].shape[axis] // parts)
            return torch.split(x[0], size, dim=axis)
        return torch.split(x[0], sizes, dim=axis)

    def _op_Slice(self, n, x, a, e):
        t = x[0]
        if len(x) == 1:                                      # opset < 10: attributes
            starts, ends, axes, steps = a["starts"], a["ends"], a.get("axes", None), None
        else:
            starts, ends = _ints(x[1]), _ints(x[2])
            axes = _ints(x[3]) if len(x) > 3 and x[3] is not None else None
            steps = _ints(x[4]) if len(x) > 4 and x[4] is not None else None
        axes = axes if axes is not None else list(range(len(starts)))
        steps = steps if steps is not None else [1] * len(starts)
        index = [slice(None)] * t.ndim
        flips = []
        for s, en, ax, st in zip(starts, ends, axes, steps):
            ax = ax if ax >= 0 else ax + t.ndim
            dim = t.shape[ax]
            if st > 0:
                index[ax] = slice(*slice(s, en, st).indices(dim))
            else:                                            # torch has no negative steps: take the mirrored range, then flip
                lo, hi, _ = slice(s, en, st).indices(dim)
                picked = list(range(lo, hi, st))
                if not picked:
                    index[ax] = slice(0, 0)
                else:
                    index[ax] = slice(picked[-1], picked[0] + 1, -st)
                    flips.append(ax)
        out = t[tuple(index)]
        return out.flip(flips) if flips else out

    def _op_Gather(self, n, x, a, e):
        import torch
        axis = a.get("axis", 0)
        data, idx = x[0], x[1]
        idx = idx.to(torch.int64)
        if data.device != idx.device:
            idx = idx.to(data.device)
        idx = torch.where(idx < 0, idx + data.shape[axis], idx)
        if idx.ndim == 0:
            return data.index_select(axis, idx.reshape(1)).squeeze(axis)
        out = data.index_select(axis, idx.reshape(-1))
        axis = axis if axis >= 0 else axis + data.ndim
        return out.reshape(*data.shape[:axis], *idx.shape, *data.shape[axis + 1:])

    def _op_Expand(self, n, x, a, e):
        import torch
        shape = _ints(x[1])
        return x[0].expand(torch.broadcast_shapes(tuple(x[0].shape), tuple(shape))).contiguous()

    def _op_Tile(self, n, x, a, e):
        return x[0].repeat(_ints(x[1]))

    def _op_Range(self, n, x, a, e):
        import torch
        return torch.arange(x[0].item(), x[1].item(), x[2].item(), dtype=x[0].dtype, device=self.device)

    def _op_Trilu(self, n, x, a, e):
        import torch
        k = int(x[1].item()) if len(x) > 1 and x[1] is not None else 0
        return torch.triu(x[0], k) if a.get("upper", 1) else torch.tril(x[0], k)

    # com.microsoft contrib operators that onnxruntime-genai's model builder (the reference's gemma3 example) and the
    # transformer optimizer emit around the MatMuls: enough of them to calibrate such a graph.  Prefill and decode alike:
    # positions and masks follow `seqlens_k` (valid length - 1 per batch entry) as onnxruntime's CPU kernels do.
    def _rms(self, t, weight, eps):
        import torch
        var = t.to(torch.float32).pow(2).mean(dim=-1, keepdim=True)
        return (t * torch.rsqrt(var + eps)).to(t.dtype) * weight

    def _op_SimplifiedLayerNormalization(self, n, x, a, e):
        axis = a.get("axis", -1)
        if axis not in (-1, x[0].ndim - 1):
            raise UnsupportedOperator("GraphRunner: SimplifiedLayerNormalization over more than the last axis")
        return (self._rms(x[0], x[1], a.get("epsilon", 1e-5)),) + (None,) * (len(n.output) - 1)

    def _op_SkipSimplifiedLayerNormalization(self, n, x, a, e):
        s = x[0] + x[1]
        if len(x) > 3 and x[3] is not None:
            s = s + x[3]
        out = (self._rms(s, x[2], a.get("epsilon", 1e-5)), None, None, s)
        return out[:max(1, len(n.output))]

    def _op_SkipLayerNormalization(self, n, x, a, e):
        import torch
        s = x[0] + x[1]
        if len(x) > 4 and x[4] is not None:
            s = s + x[4]
        y = torch.nn.functional.layer_norm(s, (s.shape[-1],), x[2], x[3] if len(x) > 3 else None, a.get("epsilon", 1e-5))
        return (y, None, None, s)[:max(1, len(n.output))]

    def _op_FastGelu(self, n, x, a, e):
        import torch
        t = x[0] + x[1] if len(x) > 1 and x[1] is not None else x[0]
        return torch.nn.functional.gelu(t, approximate="tanh")

    def _op_BiasGelu(self, n, x, a, e):
        import torch
        return torch.nn.functional.gelu(x[0] + x[1])

    def _op_QuickGelu(self, n, x, a, e):
        import torch
        return x[0] * torch.sigmoid(a.get("alpha", 1.702) * x[0])

    @staticmethod
    def _rotate(t, cos, sin, interleaved):
        """t [B, H, S, D]; cos / sin [B, S, R/2] for the first R of the D channels."""
        import torch
        r = cos.shape[-1] * 2
        rot, rest = t[..., :r], t[..., r:]
        c, s_ = cos[:, None, :, :], sin[:, None, :, :]
        if interleaved:
            x1, x2 = rot[..., 0::2], rot[..., 1::2]
            out = torch.stack((x1 * c - x2 * s_, x2 * c + x1 * s_), dim=-1).flatten(-2)
        else:
            x1, x2 = rot[..., : r // 2], rot[..., r // 2:]
            out = torch.cat((x1 * c - x2 * s_, x2 * c + x1 * s_), dim=-1)
        return torch.cat((out, rest), dim=-1) if rest.shape[-1] else out

    def _op_RotaryEmbedding(self, n, x, a, e):
        """com.microsoft::RotaryEmbedding(input [B, S, H*D] | [B, H, S, D], position_ids [B, S] | [1], cos_cache, sin_cache)."""
        import torch
        t, pos, cos_c, sin_c = x[0], x[1].to(torch.int64), x[2], x[3]
        packed = t.ndim == 3
        if packed:
            heads = a.get("num_heads", 0)
            d = cos_c.shape[-1] * 2 if not heads else t.shape[-1] // heads
            heads = heads or t.shape[-1] // d
            t = t.reshape(t.shape[0], t.shape[1], heads, d).transpose(1, 2)
        b, s_len = t.shape[0], t.shape[2]
        if pos.numel() == 1:
            pos = pos.reshape(1, 1) + torch.arange(s_len, device=pos.device).reshape(1, -1)
        pos = pos.expand(b, s_len).to(cos_c.device)
        out = self._rotate(t, cos_c[pos], sin_c[pos], bool(a.get("interleaved", 0)))
        if a.get("scale", 1.0) != 1.0:
            out = out * a.get("scale", 1.0)
        return out.transpose(1, 2).reshape(b, s_len, -1) if packed else out

    def _op_GroupQueryAttention(self, n, x, a, e):
        """com.microsoft::GroupQueryAttention: query (or packed QKV), key, value, past_key, past_value [B, Hkv, P, D], seqlens_k
        [B] (valid total length - 1), total_sequence_length, cos_cache, sin_cache -> output, present_key, present_value."""
        import torch
        x = list(x) + [None] * (9 - len(x))
        q, k, v, past_k, past_v, seqlens, _total, cos_c, sin_c = x[:9]
        if any(t is not None for t in x[9:]):
            raise UnsupportedOperator("GraphRunner: GroupQueryAttention with position_ids / attention_bias inputs")
        heads, kv_heads = a["num_heads"], a["kv_num_heads"]
        b, s_len = q.shape[0], q.shape[1]
        if k is None:                                          # packed QKV
            d = q.shape[-1] // (heads + 2 * kv_heads)
            q, k, v = torch.split(q, [heads * d, kv_heads * d, kv_heads * d], dim=-1)
        d = q.shape[-1] // heads
        q = q.reshape(b, s_len, heads, d).transpose(1, 2)
        k = k.reshape(b, s_len, kv_heads, d).transpose(1, 2)
        v = v.reshape(b, s_len, kv_heads, d).transpose(1, 2)
        dev = q.device
        past_len = 0 if past_k is None else past_k.shape[2]       # the past holds exactly the tokens seen so far (no shared buffer)
        pos = (past_len + torch.arange(s_len, device=dev)).reshape(1, -1).expand(b, s_len)     # absolute positions of the new tokens
        if a.get("do_rotary", 0):
            inter = bool(a.get("rotary_interleaved", 0))
            cos, sin = cos_c.to(dev)[pos], sin_c.to(dev)[pos]
            q, k = self._rotate(q, cos, sin, inter), self._rotate(k, cos, sin, inter)
        if past_len:
            k = torch.cat((past_k.to(k.dtype), k), dim=2)
            v = torch.cat((past_v.to(v.dtype), v), dim=2)
        total = past_len + s_len
        key_pos = torch.arange(total, device=dev).reshape(1, 1, -1)
        qpos = pos.reshape(b, s_len, 1)
        allowed = key_pos <= qpos                                 # causal
        window = a.get("local_window_size", -1)
        if window is not None and window > 0:
            allowed = allowed & (key_pos >= qpos - window)        # the token itself and `window` tokens to its left
        if seqlens is not None:                                   # right-padded prompts: keys past a row's valid length
            valid = seqlens.to(dev).to(torch.int64).reshape(b, 1, 1) + 1
            allowed = allowed & (key_pos < valid)
        scale = a.get("scale", 0.0) or 1.0 / (d ** 0.5)
        rep = heads // kv_heads
        kk, vv = k.repeat_interleave(rep, dim=1), v.repeat_interleave(rep, dim=1)
        scores = torch.matmul(q, kk.transpose(-1, -2)) * scale
        softcap = a.get("softcap", 0.0)
        if softcap:
            scores = softcap * torch.tanh(scores / softcap)
        scores = scores.masked_fill(~allowed[:, None, :, :], float("-inf"))
        probs = torch.softmax(scores, dim=-1)
        probs = torch.nan_to_num(probs)                           # rows of padding queries have no key at all
        out = torch.matmul(probs, vv).transpose(1, 2).reshape(b, s_len, heads * d)
        return (out, k, v)[:max(1, len(n.output))]

    def _op_MultiHeadAttention(self, n, x, a, e):
        """com.microsoft::MultiHeadAttention in the form the genai builder writes for models without grouped heads: query, key,
        value [B, S, N D] (three separate tensors), bias [3 N D], -, attention_bias [B | 1, N | 1, S, T] (added to the scores),
        past_key, past_value [B, N, P, D] -> output, present_key, present_value.  `unidirectional`: causal over the absolute
        positions.  Packed QKV / KV, key_padding_mask and the cache-indirection inputs are refused."""
        import torch
        x = list(x) + [None] * (8 - len(x))
        q, k, v, bias, padding, attn_bias, past_k, past_v = x[:8]
        if k is None or v is None or q.ndim != 3 or k.ndim != 3 or padding is not None or any(t is not None for t in x[8:]):
            raise UnsupportedOperator(f"GraphRunner: MultiHeadAttention (node '{n.name}') with packed inputs, key_padding_mask or cache indirection")
        heads = a["num_heads"]
        if bias is not None:
            bq, bk, bv = torch.split(bias, [q.shape[-1], k.shape[-1], v.shape[-1]])
            q, k, v = q + bq, k + bk, v + bv
        b, s_len, t_len = q.shape[0], q.shape[1], k.shape[1]
        d, dv = q.shape[-1] // heads, v.shape[-1] // heads
        q = q.reshape(b, s_len, heads, d).transpose(1, 2)
        k = k.reshape(b, t_len, heads, d).transpose(1, 2)
        v = v.reshape(b, t_len, heads, dv).transpose(1, 2)
        if past_k is not None:
            k, v = torch.cat((past_k.to(k.dtype), k), dim=2), torch.cat((past_v.to(v.dtype), v), dim=2)
        total = k.shape[2]
        scale = a.get("scale", 0.0) or 1.0 / (d ** 0.5)
        scores = torch.matmul(q, k.transpose(-1, -2)) * scale
        if attn_bias is not None:
            scores = scores + attn_bias
        if a.get("unidirectional", 0):
            qpos = (total - s_len + torch.arange(s_len, device=q.device)).reshape(-1, 1)
            scores = scores.masked_fill(torch.arange(total, device=q.device).reshape(1, -1) > qpos, float("-inf"))
        out = torch.matmul(torch.softmax(scores, dim=-1), v).transpose(1, 2).reshape(b, s_len, heads * dv)
        return (out, k, v)[:max(1, len(n.output))]

    # quantization operators (ONNX QuantizeLinear / DequantizeLinear / DynamicQuantizeLinear, opset 21 semantics)
    @staticmethod
    def _per_axis(p, like, axis):
        if p.ndim == 1 and like.ndim > 1:
            shape = [1] * like.ndim
            shape[axis] = -1
            return p.reshape(shape)
        return p

    def _op_DequantizeLinear(self, n, x, a, e):
        import torch
        q, scale = x[0], x[1]
        zp = x[2] if len(x) > 2 and x[2] is not None else None
        axis, block = a.get("axis", 1), a.get("block_size", 0)
        axis = axis if axis >= 0 else axis + q.ndim
        qf = q.to(torch.float32)
        if block:                                            # blocked along `axis`: parameters repeat `block` times
            scale = scale.repeat_interleave(block, dim=axis)
            zpf = zp.to(torch.float32).repeat_interleave(block, dim=axis) if zp is not None else None
            return ((qf - zpf) if zpf is not None else qf) * scale
        scale = self._per_axis(scale, q, axis)
        if zp is not None:
            qf = qf - self._per_axis(zp.to(torch.float32), q, axis)
        return qf * scale

    def _quantize(self, t, scale, zp, code=None):
        import torch
        dt = zp.dtype if zp is not None else torch.uint8
        lo, hi = (0, 255) if dt == torch.uint8 else (-128, 127)
        q = torch.round(t / scale)                           # half to even, like the operator
        if zp is not None:
            q = q + zp.to(torch.float32)
        return q.clamp(lo, hi).to(dt)

    def _op_QuantizeLinear(self, n, x, a, e):
        zp = x[2] if len(x) > 2 and x[2] is not None else None
        axis = a.get("axis", 1)
        axis = axis if axis >= 0 else axis + x[0].ndim
        if a.get("block_size", 0):
            raise UnsupportedOperator("GraphRunner: blocked QuantizeLinear")
        scale = self._per_axis(x[1], x[0], axis)
        return self._quantize(x[0], scale, None if zp is None else self._per_axis(zp, x[0], axis).to(zp.dtype))

    def _op_DynamicQuantizeLinear(self, n, x, a, e):
        import torch
        t = x[0]
        lo = torch.clamp(t.min(), max=0.0)
        hi = torch.clamp(t.max(), min=0.0)
        scale = (hi - lo) / 255.0
        safe = torch.where(scale == 0, torch.ones_like(scale), scale)
        zp = torch.round(torch.clamp(-lo / safe, 0, 255)).to(torch.uint8)
        q = torch.clamp(torch.round(t / safe) + zp.to(torch.float32), 0, 255).to(torch.uint8)
        return q, scale, zp

    def _op_QLinearMatMul(self, n, x, a, e):
        import torch
        A = (x[0].to(torch.float32) - x[2].to(torch.float32)) * x[1]
        bz = self._per_axis(x[5].to(torch.float32), x[3], x[3].ndim - 1)
        B = (x[3].to(torch.float32) - bz) * self._per_axis(x[4], x[3], x[3].ndim - 1)
        return self._quantize(A @ B, x[6], x[7])

    def _op_QGemm(self, n, x, a, e):
        """com.microsoft::QGemm: A, a_scale, a_zp, B, b_scale, b_zp, C (int32, scale a_scale * b_scale), y_scale, y_zp."""
        import torch
        A = (x[0].to(torch.float32) - x[2].to(torch.float32)) * x[1]
        if a.get("transA", 0):
            A = A.t()
        Bq = x[3].t() if a.get("transB", 0) else x[3]
        bs, bz = x[4], x[5].to(torch.float32)
        B = (Bq.to(torch.float32) - (bz if bz.ndim == 0 else bz.reshape(1, -1))) * (bs if bs.ndim == 0 else bs.reshape(1, -1))
        out = (A @ B) * a.get("alpha", 1.0)
        if len(x) > 6 and x[6] is not None:
            out = out + x[6].to(torch.float32) * (x[1] * (bs if bs.ndim == 0 else bs.reshape(1, -1)))
        if len(x) > 7 and x[7] is not None:
            return self._quantize(out, x[7], x[8] if len(x) > 8 else None)
        return out

    def _op_MatMulNBits(self, n, x, a, e):
        """com.microsoft::MatMulNBits: B [N, K/g, g*bits/8] (4-bit: low nibble first), scales [N, K/g] (or flat),
        zero points packed like B along the block axis (default 2^(bits-1)), optional g_idx (unused by the writer), bias."""
        import torch
        K, N, bits, g = a["K"], a["N"], a.get("bits", 4), a["block_size"]
        blocks = (K + g - 1) // g
        blob = x[1].reshape(N, blocks, -1).to(torch.uint8)
        if bits == 4:
            q = torch.stack([blob & 0x0F, blob >> 4], dim=-1).reshape(N, blocks, g)
        elif bits == 8:
            q = blob.reshape(N, blocks, g)
        else:
            raise UnsupportedOperator(f"GraphRunner: MatMulNBits with bits = {bits}")
        scales = x[2].reshape(N, blocks).to(torch.float32)
        zp = x[3] if len(x) > 3 and x[3] is not None else None
        if zp is None:
            zpf = torch.full((N, blocks), float(1 << (bits - 1)), device=q.device)
        elif zp.is_floating_point():
            zpf = zp.reshape(N, blocks).to(torch.float32)
        elif bits == 4:
            z = zp.reshape(N, -1).to(torch.uint8)
            zpf = torch.stack([z & 0x0F, z >> 4], dim=-1).reshape(N, -1)[:, :blocks].to(torch.float32)
        else:
            zpf = zp.reshape(N, blocks).to(torch.float32)
        if len(x) > 4 and x[4] is not None:
            raise UnsupportedOperator("GraphRunner: MatMulNBits with g_idx")
        w = ((q.to(torch.float32) - zpf[:, :, None]) * scales[:, :, None]).reshape(N, blocks * g)[:, :K]
        out = x[0] @ w.t()
        if len(x) > 5 and x[5] is not None:
            out = out + x[5]
        return out


_CAPTURE_MAX_INPUT_ELEMENTS = 1 << 20      # recording pays when a pass is hundreds of small launches (one short sequence), not above
_PIECES_MIN_WIDTH, _PIECES_MIN_ROWS = 512, 256
_PIECES_HEADROOM = 8                      # free HBM, in sizes of the weight, below which its fp16 pieces are not kept
_SCALAR_FRIENDLY = {"Add", "Sub", "Mul", "Div", "Pow"}
_HOST_OPERANDS = {"GroupQueryAttention", "RotaryEmbedding", "CumSum", "Reshape", "Expand", "Slice", "Tile", "Unsqueeze", "Squeeze", "Split",
                  "ConstantOfShape", "Gather", "Trilu", "ReduceMean", "ReduceSum", "ReduceMax", "ReduceMin", "Range", "Clip", "Pad", "Resize", "TopK", "OneHot"}


class _Attrs:
    """Attribute access with defaults; values decoded on first use."""

    def __init__(self, protos):
        self._p = protos

    def __contains__(self, name):
        return name in self._p

    def __getitem__(self, name):
        return attribute_value(self._p[name])

    def get(self, name, default=None):
        return attribute_value(self._p[name]) if name in self._p else default

