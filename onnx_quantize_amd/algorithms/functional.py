"""NumPy-facing mirrors of the reference's numeric helpers (core/_algorithms/utils.py), computed on the
GPU through the C ABI.  Same names, argument order and result shapes / dtypes as the reference so that
callers (and tests written against the reference) can switch by changing the import.

Host arrays are copied to HBM, processed, and copied back (the reference's contract is NumPy in, NumPy
out); pipelines that already hold their tensors in HBM should call ``onnx_quantize_amd.hip.ops``
directly.  The HIP path computes in fp32: float64 inputs are rounded to fp32 first.  There is no CPU
fallback.
"""
from __future__ import annotations

import numpy as np

from ..config import QuantizationStrategy
from ..dtypes import QuantType

__all__ = [
    "_preprocess_array", "_post_process_array", "_compute_min_max", "_compute_qparams",
    "_compute_qparams_from_array", "_quantize_array_from_qparams", "_dequantize_array",
    "_fake_quantize_array",
]


def _dev(a, dtype=np.float32):
    import torch

    return torch.from_numpy(np.ascontiguousarray(np.asarray(a), dtype=dtype)).cuda()


def _sname(strategy) -> str:
    assert isinstance(strategy, QuantizationStrategy)
    return strategy.value


# ------------------------------------------------------------------ L1 / L2: pure views, no arithmetic
def _preprocess_array(array, strategy, group_size=-1):
    """utils.py:6-26."""
    assert isinstance(strategy, QuantizationStrategy)
    if strategy == QuantizationStrategy.TENSOR:
        return array
    if strategy == QuantizationStrategy.CHANNEL:
        return array.T
    in_channels = array.shape[0]
    g = min(group_size, in_channels)
    g = g if g != -1 else in_channels
    return array.T.reshape((-1, g))


def _post_process_array(preprocessed_array, original_array, strategy, group_size=-1):
    """utils.py:29-39."""
    assert isinstance(strategy, QuantizationStrategy)
    if strategy == QuantizationStrategy.TENSOR:
        return preprocessed_array
    if strategy == QuantizationStrategy.CHANNEL:
        return preprocessed_array.T
    return preprocessed_array.reshape(original_array.T.shape).T


def _rows_as_kn(rows):
    """A row-layout array (rows share parameters) as the [K, N] matrix whose columns are those rows."""
    rows = np.asarray(rows)
    if rows.ndim == 1:
        rows = rows[None, :]
    return np.ascontiguousarray(rows.reshape(rows.shape[0], -1).T, dtype=np.float32)


# ------------------------------------------------------------------ R1
def _tensor_extrema(a: np.ndarray):
    """Global (min, max) of a float32 / float64 array through the calibration reduction kernel."""
    import torch

    from ..hip import ops

    dt = np.float64 if a.dtype == np.float64 else np.float32
    x = _dev(a, dt)
    st = ops.minmax_state(x.device, torch.float64 if dt == np.float64 else torch.float32)
    ops.minmax_collect(x, st, 0.0)
    lo, hi = st[:2].tolist()
    return dt(lo), dt(hi)


def _compute_min_max(array, strategy, group_size=-1, clip_ratio=1.0):
    """utils.py:42-69 (per row of the preprocessed layout for channel / group)."""
    from ..hip import ops

    assert isinstance(strategy, QuantizationStrategy)
    a = np.asarray(array)
    if strategy == QuantizationStrategy.TENSOR:
        lo, hi = _tensor_extrema(a)
        lo, hi = lo * clip_ratio, hi * clip_ratio            # NumPy scalar * Python float keeps the dtype (NEP 50)
        return np.array(np.minimum(lo, 0)), np.array(np.maximum(hi, 0))
    rows = a.reshape(a.shape[0], -1) if a.ndim != 2 else a
    mn, mx = ops.minmax_rows(_dev(rows))
    lo = mn.cpu().numpy().reshape(-1, 1) * np.float32(clip_ratio)
    hi = mx.cpu().numpy().reshape(-1, 1) * np.float32(clip_ratio)
    return np.minimum(lo, 0), np.maximum(hi, 0)


# ------------------------------------------------------------------ Q1
def _compute_qparams(rmin, rmax, quant_type, is_symmetric, reduce_range, scale_dtype, zp_dtype):
    """utils.py:242-299.  float64 ranges are evaluated in double like NumPy does, everything else in fp32."""
    from ..hip import ops

    rmin, rmax = np.asarray(rmin), np.asarray(rmax)
    if rmin.dtype == np.float64 or rmax.dtype == np.float64:
        s, z = ops.qparams_f64(_dev(rmin, np.float64).reshape(rmin.shape), _dev(rmax, np.float64).reshape(rmax.shape),
                               quant_type.key, bool(is_symmetric), bool(reduce_range))
    else:
        s, z = ops.qparams(_dev(rmin).reshape(rmin.shape), _dev(rmax).reshape(rmax.shape), quant_type.key,
                           bool(is_symmetric), bool(reduce_range))
    scale = s.cpu().numpy().reshape(rmin.shape).astype(scale_dtype)
    zp = np.asarray(z.cpu().numpy().reshape(rmin.shape), dtype=zp_dtype)
    return scale, zp


# ------------------------------------------------------------------ Q2
def _compute_qparams_from_array(array, quant_type, strategy, group_size, is_symmetric, reduce_range,
                                clip_ratio, mse, scale_dtype, zp_dtype):
    """utils.py:302-348.  ``array`` is in the reference's preprocessed layout: every row shares one
    (scale, zero point) for channel / group, the whole array for tensor."""
    from ..hip import ops

    sname = _sname(strategy)
    if sname == "tensor" and np.asarray(array).dtype == np.float64 and not mse:
        # the reference's arithmetic follows the input dtype; keep float64 inputs in double (its own KATs do this)
        lo, hi = _compute_min_max(array, strategy, group_size, clip_ratio)
        return _compute_qparams(lo, hi, quant_type, is_symmetric, reduce_range, scale_dtype, zp_dtype)
    if sname == "tensor":
        a = np.asarray(array, dtype=np.float32)
        a2 = a.reshape(1, -1) if a.ndim != 2 else a
        _, s, z = ops.rtn_quantize(_dev(a2), quant_type.key, "tensor", -1, is_symmetric, reduce_range,
                                   clip_ratio, mse, emit_q=False)
        scale = s.cpu().numpy().astype(scale_dtype)
        zp = np.asarray(z.cpu().numpy(), dtype=zp_dtype)
        return scale, zp
    kn = _rows_as_kn(array)                                   # [row length, number of rows]
    _, s, z = ops.rtn_quantize(_dev(kn), quant_type.key, "channel", -1, is_symmetric, reduce_range,
                               clip_ratio, mse, emit_q=False)
    scale = s.cpu().numpy().reshape(-1, 1).astype(scale_dtype)
    zp = np.asarray(z.cpu().numpy().reshape(-1, 1), dtype=zp_dtype)
    return scale, zp


# ------------------------------------------------------------------ K1 - K3
def _param_mode(x: np.ndarray, scale) -> str:
    n = np.size(scale)
    if n == 1:
        return "tensor"
    if x.ndim == 2 and n == x.shape[0]:
        return "row"
    if n == x.shape[-1]:
        return "col"
    raise ValueError(f"cannot broadcast {np.shape(scale)} parameters over an array of shape {x.shape}")


def _quantize_array_from_qparams(array, scale, zero_point, quant_type, is_symmetric, reduce_range):
    """utils.py:72-79.  ``scale`` / ``zero_point`` broadcast like NumPy: scalars, [R, 1] per row, or [C]."""
    from ..hip import ops

    x = np.asarray(array, dtype=np.float32)
    mode = _param_mode(x, scale)
    q = ops.quantize(_dev(x), _dev(scale), _dev(np.asarray(zero_point), np.int32), quant_type.key,
                     bool(is_symmetric), bool(reduce_range), mode=mode)
    return q.cpu().numpy().astype(quant_type.np_dtype, copy=False)


def _dequantize_array(q_array, scale, zero_point, *, preprocess=False, strategy=None, group_size=-1):
    """utils.py:102-137."""
    from ..hip import ops

    q = np.asarray(q_array)
    kind = {"int8": "int8", "uint8": "uint8", "int32": "int32", "uint32": "uint32"}.get(q.dtype.name)
    if kind is None:   # ml_dtypes 4-bit containers hold one value per byte
        kind = "int8" if np.issubdtype(q.dtype, np.signedinteger) or "int4" == q.dtype.name else "uint8"
        q = q.astype(np.int8 if kind == "int8" else np.uint8)
    s = np.asarray(scale, dtype=np.float32)
    z = np.asarray(zero_point)
    import torch

    qd = torch.from_numpy(np.ascontiguousarray(q)).cuda()
    zdt = np.float32 if z.dtype.kind == "f" else np.int32          # HQQ's float zero points stay floats (utils.py:131)
    if preprocess:
        assert strategy is not None, "strategy must be provided if preprocess is True"
        sname = _sname(strategy)
        if sname == "tensor":
            out = ops.dequantize(qd, _dev(s), _dev(z, zdt), kind, mode="tensor")
        elif sname == "channel":
            out = ops.dequantize(qd, _dev(s), _dev(z, zdt), kind, mode="col")
        else:
            k = q.shape[0]
            g = min(group_size, k)
            g = g if g != -1 else k
            out = ops.dequantize(qd, _dev(s), _dev(z, zdt), kind, mode="group", group=g)   # ragged groups: flat addressing
        return out.cpu().numpy()
    mode = _param_mode(q, s)
    return ops.dequantize(qd, _dev(s), _dev(z, zdt), kind, mode=mode).cpu().numpy()


def _fake_quantize_array(array, scale, zero_point, quant_type, is_symmetric, reduce_range):
    """utils.py:82-99."""
    q = _quantize_array_from_qparams(array, scale, zero_point, quant_type, is_symmetric, reduce_range)
    return _dequantize_array(q, scale, zero_point)
