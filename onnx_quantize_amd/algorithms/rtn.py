"""Round-to-nearest weight quantization on the GPU (reference: core/_algorithms/rtn.py)."""
from __future__ import annotations

from typing import TYPE_CHECKING, Literal

import numpy as np

from ..config import AlgorithmConfig, QuantizationStrategy, register_algorithm_config
from ..dtypes import QuantType

if TYPE_CHECKING:  # pragma: no cover
    import onnx_ir as ir

    from ..config import QConfig

__all__ = ["RTNConfig", "_rtn_quantize"]


@register_algorithm_config
class RTNConfig(AlgorithmConfig):
    """Default algorithm; no parameters of its own (rtn.py:28-51)."""

    algorithm_type: Literal["rtn"] = "rtn"

    def quantize_weights(self, w: "ir.Value", qconfig: "QConfig", out: "ir.Value | None" = None):
        a = qconfig.weights
        return _rtn_quantize(w.const_value.numpy(), a.dtype, strategy=a.strategy, group_size=a.group_size,
                             is_symmetric=a.symmetric, reduce_range=a.reduce_range, clip_ratio=a.clip_ratio,
                             mse=a.mse, scale_dtype=a.scale_dtype, zp_dtype=a.zp_dtype)


def _rtn_quantize(array: np.ndarray, quant_type: QuantType, strategy: QuantizationStrategy, group_size: int,
                  is_symmetric: bool, reduce_range: bool, clip_ratio: float, mse: bool, scale_dtype: np.dtype,
                  zp_dtype: np.dtype) -> tuple[np.ndarray, np.ndarray, np.ndarray]:
    """rtn.py:54-109: one fused HIP kernel launch (oq_rtn_quantize_f32) instead of the reference's
    transpose-copy + min/max + divide/round/clip NumPy passes.  ``array`` is [K, N]; the result is
    (q [K, N], scale, zero_point) with the reference's shapes: 0-d | [N] | [N*K/g, 1]."""
    import torch

    from ..hip import ops

    assert isinstance(strategy, QuantizationStrategy)
    w = np.asarray(array)
    if w.ndim != 2:
        w = w.reshape(1, -1) if strategy == QuantizationStrategy.TENSOR else w.reshape(w.shape[0], -1)
    from ..staging import download, upload

    wd = upload(w)

    q, s, z = ops.rtn_quantize(wd, quant_type.key, strategy.value, -1 if group_size is None else group_size,
                               bool(is_symmetric), bool(reduce_range), float(clip_ratio), bool(mse))
    q_np = download(q).reshape(np.shape(array)).astype(quant_type.np_dtype, copy=False)
    scale = s.cpu().numpy().astype(scale_dtype, copy=False)
    zp = z.cpu().numpy().astype(zp_dtype, copy=False)
    return q_np, scale, zp


def _quantize_bias(bias, input_scale, weight_scale):
    """rtn.py:112-138: int32 bias, scale = weight_scale * input_scale, zero point 0."""
    import torch

    from ..hip import ops

    assert bias.ndim == 1
    assert bias.dtype == np.float32
    assert np.size(input_scale) == 1
    assert weight_scale.dtype == np.float32
    assert weight_scale.size == 1 or bias.size == weight_scale.size
    q, bs = ops.quantize_bias(torch.from_numpy(np.ascontiguousarray(bias)).cuda(), float(np.float32(input_scale)),
                              torch.from_numpy(np.ascontiguousarray(weight_scale).reshape(-1)).cuda())
    bias_scale = bs.cpu().numpy()
    if weight_scale.size == 1:
        bias_scale = bias_scale[:1].reshape(np.shape(weight_scale))
    return q.cpu().numpy(), bias_scale, 0
