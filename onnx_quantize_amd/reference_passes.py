"""The reference's own pass and calibration entry points with their numeric cores on the GPU.

`integration.install_into_reference()` puts the functions below in place of

* ``AwqPass._apply_awq`` / ``_apply_awq_clip``        (pre_passes/awq.py:114-204, :206-259)
* ``SmoothQuantPass._smooth_quant_node``             (pre_passes/smooth_quant.py:91-134)
* ``calibrate._set_qparams`` / ``_set_qparams_gptq`` (core/_calibration/calibrate.py:254-285, :288-307) and a wrapper
  around ``calibrate_model`` (:310-385) that tells them what the run will ask for.

Until round 3 only the two helpers the AWQ pass calls were rebound, so a user of ``quantize()`` with ``AwqConfig`` ran the
reference's 20-iteration Python loop with one upload + download of the whole weight per candidate and `np.matmul` on the
host, and the calibrator was fed one tensor per call.  Here every method keeps the reference's signature, its checks and
its GRAPH EDITS (the statements are the reference's: same initializer names, same `node.meta` updates, same order), and
replaces only the arithmetic in between by one call into `hip.ops`:

    _apply_awq        upload X and W once -> oq_awq_scale_search_f32 (statistics, 20 candidates, RTN, X (W - W^) on the
                      matrix cores, argmin on the device) -> the winning scale [K] comes back
    _apply_awq_clip   upload X and W once -> oq_awq_clip_search_f32 (10 clip ratios), the winning ratio comes back
    _smooth_quant_node -> oq_smooth_quant_scale_f32
    _set_qparams      every batch dict uploaded once -> `MinMaxCalibrator.collect_many` (one launch pair per batch); when
                      both activation kinds are calibrated the second walk is replayed from per-batch extrema kept on
                      the device instead of uploading the activations again (`calibration_driver.ActivationStream`)
    _set_qparams_gptq the Hessian of every tapped input is accumulated batch by batch on the device and
                      `node.meta["input"]` becomes a `StreamedGptqInput` (H, n) -- unless a preprocessor (AWQ,
                      SmoothQuant) needs the activations themselves: then the reference's concatenation is kept, so
                      that `node.meta["input"]` stays the ndarray those passes read and rescale in place.

Nothing here computes on the CPU: the device functions raise when the HIP library or a GPU is missing.
"""
from __future__ import annotations

import contextvars
import logging

import numpy as np

__all__ = ["StreamedGptqInput", "install_awq", "install_smooth_quant", "install_calibrate"]

logger = logging.getLogger(__name__)


def _dev(a):
    """One blocking upload (fp32, C-contiguous) -- `staging.upload`; a module-level name so that the CPU tests of the graph
    edits can stand in for the device."""
    from .staging import upload

    return upload(np.asarray(a))


def _ops():
    from .hip import ops

    return ops


def _key(qtype) -> str:
    return qtype.name[1:].lower()          # "QUInt4" -> "uint4" for this package's QuantType and the reference's alike


def _our_qtype(qtype):
    from .dtypes import QuantType

    return qtype if isinstance(qtype, QuantType) else QuantType[qtype.name]


class _Args:
    """The three fields of QActivationArgs the stream reads, with this package's QuantType (the reference's enum has the
    same member names but is another class)."""

    def __init__(self, qargs):
        self.dtype, self.symmetric, self.reduce_range = _our_qtype(qargs.dtype), bool(qargs.symmetric), bool(qargs.reduce_range)


def _strategy(s) -> str:
    return s.value if hasattr(s, "value") else str(s)


class StreamedGptqInput:
    """What `_set_qparams_gptq` leaves in ``node.meta["input"]`` when only the weight algorithm needs the activations: the
    Hessian H = (2 / n) sum_b X_b^T X_b of the value (gptq.py:246-260 telescoped over the batches, `HessianAccumulator`) and
    the sample count, both already on the device.  Nodes that read the same value share ONE object, as they share one
    array in the reference (calibrate.py:301-307); the seam factors it once (`seam._hessian_and_factor`)."""

    __slots__ = ("name", "h", "n", "shape", "factors")

    def __init__(self, name, h, n, shape):
        self.name, self.h, self.n, self.shape = name, h, int(n), tuple(shape)
        self.factors = {}        # (percdamp, actorder, Hessian method, device) -> what `_gptq` derives from H alone (seam.prefactor_streamed)

    def __repr__(self):  # pragma: no cover
        return f"StreamedGptqInput({self.name!r}, K={self.h.shape[0]}, n={self.n})"


# --------------------------------------------------------------------------------------------- AWQ / SmoothQuant
def install_awq(ref_awq) -> None:
    """Rebind the two search methods of ``ref_awq.AwqPass`` (the module is passed in: its `ir` and `QConfig` are the
    ones the pass was written against)."""
    ir, QConfig = ref_awq.ir, ref_awq.QConfig

    def _apply_awq(self, node, model) -> bool:
        if not self.is_valid_node(node):
            return False
        qconfig = QConfig(**node.meta["qconfig"])
        a = qconfig.weights
        original_weights = ir.convenience.get_const_tensor(node.inputs[1]).numpy()
        # 1.-3. activation / weight statistics, the 20 candidate scales and their losses: one device call
        x_dev, w_dev = _dev(node.meta["input"]), _dev(original_weights)
        best_dev, _losses = _ops().awq_scale_search(x_dev, w_dev, _key(a.dtype), _strategy(a.strategy), a.group_size,
                                                    bool(a.symmetric), bool(a.reduce_range))
        best_scale = _host(best_dev)
        # 4. Fuse scale into weights
        updated_weights = original_weights * best_scale.reshape(-1, 1)
        # 5. Create a Mul node to scale the input activations
        scale_initializer = ir.val(f"{node.outputs[0].name}_scale", const_value=ir.tensor(1.0 / best_scale))
        # 6. Input activation metadata need to be updated as the activations were scaled
        node.meta["input"] /= best_scale.reshape((1, -1))
        # 7. Insert the Mul node before the current node
        self._insert_mul_node_before(node, model, scale_initializer)
        # 8. Update weight initializer
        weights_initializer = ir.val(node.inputs[1].name, const_value=ir.tensor(updated_weights))
        ir.convenience.replace_all_uses_with(node.inputs[1], weights_initializer)
        model.graph.initializers[node.inputs[1].name] = weights_initializer
        return True

    def _apply_awq_clip(self, node) -> bool:
        if not self.is_valid_node(node):
            return False
        qconfig = QConfig(**node.meta["qconfig"])
        a = qconfig.weights
        x_dev, w_dev = _dev(node.meta["input"]), _dev(ir.convenience.get_const_tensor(node.inputs[1]).numpy())
        best_ratio, _losses = _ops().awq_clip_search(x_dev, w_dev, _key(a.dtype), _strategy(a.strategy), a.group_size,
                                                     bool(a.symmetric), bool(a.reduce_range))
        qconfig.weights.clip_ratio = best_ratio
        node.meta["qconfig"] = qconfig.model_dump()
        return True

    _apply_awq.__doc__ = "pre_passes/awq.py:114-204 with steps 1-3 (statistics, grid, losses) on the GPU"
    _apply_awq_clip.__doc__ = "pre_passes/awq.py:206-259 with the 10-point search on the GPU"
    for fn in (_apply_awq, _apply_awq_clip):
        fn._oq_rebound = True
    ref_awq.AwqPass._apply_awq = _apply_awq
    ref_awq.AwqPass._apply_awq_clip = _apply_awq_clip


def _host(t) -> np.ndarray:
    if isinstance(t, np.ndarray):
        return t
    from .staging import download

    return download(t)


def install_smooth_quant(ref_sq) -> None:
    ir, QConfig = ref_sq.ir, ref_sq.QConfig

    def _smooth_quant_node(self, node, model) -> bool:
        if node.op_type not in self.target_op_types or node.domain != "":
            return False
        if ir.convenience.get_const_tensor(node.inputs[1]) is None:
            return False
        if node.meta.get("qconfig") is None:
            return False
        qconfig = QConfig(**node.meta["qconfig"])
        if not qconfig.preprocessors:
            return False
        # 1.-3. activation absmax, weight absmax and the smoothing scale: one device call
        weights = ir.convenience.get_const_tensor(node.inputs[1]).numpy()
        scale = _host(_ops().smooth_quant_scale(_dev(node.meta["input"]), _dev(weights), float(self.alpha)))
        # 4. Fuse scale into weights
        updated_weights = np.multiply(scale.reshape(-1, 1), weights)
        # 5. Create a Mul node to scale the input activations
        scale_initializer = ir.val(f"{node.outputs[0].name}_scale", const_value=ir.tensor(1.0 / scale))
        # 6. Input activation metadata need to be updated as the activations were scaled
        node.meta["input"] /= scale.reshape((1, -1))
        # 7. Insert the Mul node before the current node
        self._insert_mul_node_before(node, model, scale_initializer)
        # 8. Update weight initializer
        weights_initializer = ir.val(node.inputs[1].name, const_value=ir.tensor(updated_weights))
        ir.convenience.replace_all_uses_with(node.inputs[1], weights_initializer)
        model.graph.initializers[node.inputs[1].name] = weights_initializer
        return True

    _smooth_quant_node.__doc__ = "pre_passes/smooth_quant.py:91-134 with steps 1-3 on the GPU"
    _smooth_quant_node._oq_rebound = True
    ref_sq.SmoothQuantPass._smooth_quant_node = _smooth_quant_node


# --------------------------------------------------------------------------------------------- calibration walk
class _Run:
    """What one `calibrate_model` call will ask of the activation list (calibrate.py:320-331), and the stream that serves
    all of it from ONE pass over the batches."""

    def __init__(self, inputs: bool, outputs: bool, algorithm: bool, preprocessing: bool):
        self.inputs, self.outputs, self.algorithm, self.preprocessing = inputs, outputs, algorithm, preprocessing
        self.stream = None
        self.fed = None            # the activation list the stream was fed from (held: its id cannot be reused meanwhile)

    @property
    def streamed_hessians(self) -> bool:
        # AWQ / SmoothQuant read and rescale node.meta["input"] as an ndarray (awq.py:121-195, smooth_quant.py:107-124)
        return self.algorithm and not self.preprocessing


_RUN: contextvars.ContextVar = contextvars.ContextVar("oq_calibration_run", default=None)


def _upload_batch(activation: dict) -> dict:
    import torch

    out = {}
    for name, data in activation.items():
        if isinstance(data, torch.Tensor):
            out[name] = data if data.is_cuda else data.cuda()
            continue
        arr = np.ascontiguousarray(data)
        if arr.dtype == np.float16:
            arr = arr.astype(np.float32)
        out[name] = torch.from_numpy(arr).cuda()
    return out


def _feed_once(run: _Run, ir_model, activations, nodes_to_calibrate, calibrator):
    """Upload every batch dict once and hand it to an `ActivationStream` that collects whatever this run needs: ranges of
    all tapped names (calibrate.py:264-266 collects every name in both walks), Hessians of the GPTQ inputs."""
    from .calibration_driver import ActivationStream

    if run.stream is not None and run.fed is activations:
        return run.stream
    in_names = [n.inputs[0].name for n in ir_model.graph if n in nodes_to_calibrate] if run.inputs else []
    out_names = [n.outputs[0].name for n in ir_model.graph if n in nodes_to_calibrate] if run.outputs else []
    hess = {n.inputs[0].name for n in ir_model.graph if n in nodes_to_calibrate} if run.streamed_hessians else set()
    stream = ActivationStream(calibrator=calibrator, input_names=in_names, output_names=out_names, hessian_names=hess)
    for activation in activations:
        stream.feed(_upload_batch(activation))
    run.stream, run.fed = stream, activations
    return stream


def install_calibrate(ref_calibrate, also=()) -> None:
    """Rebind `_set_qparams`, `_set_qparams_gptq` and wrap `calibrate_model` in ``ref_calibrate``; ``also`` lists modules
    that imported `calibrate_model` by name (pre_passes/__init__.py:8)."""
    if getattr(ref_calibrate.calibrate_model, "_oq_rebound", False):
        return
    original_set_qparams = ref_calibrate._set_qparams
    original_set_qparams_gptq = ref_calibrate._set_qparams_gptq
    original_calibrate_model = ref_calibrate.calibrate_model
    Kind = ref_calibrate._ActivationKind

    def _set_qparams(ir_model, activations, nodes_to_calibrate, calibrator, qargs, kind) -> None:
        if not hasattr(calibrator, "collect_many"):          # a calibrator plugin of the user's: the reference's own walk
            return original_set_qparams(ir_model, activations, nodes_to_calibrate, calibrator, qargs, kind)
        run = _RUN.get()
        if run is not None and ((kind == Kind.INPUT and run.inputs) or (kind == Kind.OUTPUT and run.outputs)):
            stream = _feed_once(run, ir_model, activations, nodes_to_calibrate, calibrator)
            qparams = stream.input_qparams(_Args(qargs)) if kind == Kind.INPUT else stream.output_qparams(_Args(qargs))
        else:   # called outside `calibrate_model`: calibrate.py:264-266 with one launch pair per batch dict
            for activation in activations:
                calibrator.collect_many(_upload_batch(activation))
            names = [n.inputs[0].name if kind == Kind.INPUT else n.outputs[0].name for n in ir_model.graph if n in nodes_to_calibrate]
            have = [n for n in dict.fromkeys(names) if n in calibrator.data]
            qparams = calibrator.compute_qparams_many(have, _our_qtype(qargs.dtype), bool(qargs.symmetric), bool(qargs.reduce_range))
        for node in ir_model.graph:
            name = node.inputs[0].name if kind == Kind.INPUT else node.outputs[0].name
            if node in nodes_to_calibrate and name in qparams:
                scale, zero_point = qparams[name]
                node.meta[f"{kind.value}_scale"] = scale.astype(qargs.scale_dtype, copy=False)
                node.meta[f"{kind.value}_zero_point"] = zero_point.astype(qargs.zp_dtype, copy=False)

    def _set_qparams_gptq(ir_model, activations, nodes_to_calibrate) -> None:
        run = _RUN.get()
        if run is None or not run.streamed_hessians:
            return original_set_qparams_gptq(ir_model, activations, nodes_to_calibrate)
        from .calibration import MinMaxCalibrator

        stream = _feed_once(run, ir_model, activations, nodes_to_calibrate, MinMaxCalibrator())
        shapes = {}
        for activation in activations:                       # leading-dimension bookkeeping only: what np.concatenate would give
            for name, data in activation.items():
                first = shapes.setdefault(name, [0, tuple(data.shape[1:])])
                first[0] += int(data.shape[0])
        shared = {}
        for node in ir_model.graph:
            name = node.inputs[0].name
            if node in nodes_to_calibrate and name in stream.hessians:
                if name not in shared:
                    acc = stream.hessians[name]
                    shared[name] = StreamedGptqInput(name, acc.h, acc.n, (shapes[name][0], *shapes[name][1]))
                node.meta["input"] = shared[name]

    def calibrate_model(ir_model, qconfig):
        run = _Run(inputs=qconfig.input_activations is not None and qconfig.input_activations.is_static,
                   outputs=qconfig.output_activations is not None and qconfig.output_activations.is_static,
                   algorithm=qconfig.weights is not None and qconfig.weights.algorithm.requires_calibration,
                   preprocessing=any(pre.requires_calibration for pre in qconfig.preprocessors))
        token = _RUN.set(run)
        try:
            return original_calibrate_model(ir_model, qconfig)
        finally:
            _RUN.reset(token)

    calibrate_model.__doc__ = original_calibrate_model.__doc__
    for fn in (_set_qparams, _set_qparams_gptq, calibrate_model):
        fn._oq_rebound = True
    calibrate_model._oq_original = original_calibrate_model
    ref_calibrate._set_qparams = _set_qparams
    ref_calibrate._set_qparams_gptq = _set_qparams_gptq
    ref_calibrate.calibrate_model = calibrate_model
    for mod in also:
        if getattr(mod, "calibrate_model", None) is original_calibrate_model:
            mod.calibrate_model = calibrate_model
