"""MatMulNBits wire format: the host-side decisions and the array preparation of the reference's
`qrules/_common.py`, the latter computed on the GPU.

* `_resolve_group_size`        qrules/_common.py:13-29  (takes the number of input channels instead of an ir.Value)
* `is_matmul_nbits_compatible` qrules/_common.py:32-62
* `_prepare_for_matmul_nbits`  qrules/_common.py:65-123  NumPy in / NumPy out like the reference, through
  `oq_pack_matmul_nbits` and `oq_pack_zero_points_u4`; pipelines that hold the weights in HBM skip it altogether:
  `ops.rtn_quantize(..., layout="nbits")` / `ops.hqq_quantize(..., layout="nbits")` write the blob directly.
"""
from __future__ import annotations

import logging

import numpy as np

from .config import QConfig, QuantizationStrategy
from .dtypes import QuantType

logger = logging.getLogger(__name__)

__all__ = ["_resolve_group_size", "is_matmul_nbits_compatible", "_prepare_for_matmul_nbits"]


def _resolve_group_size(in_channels: int, group_size, name: str = "") -> int:
    """_common.py:13-29: a group size that exceeds or does not divide the input channels becomes the input channels."""
    msg = f"Adjusting group size from {group_size} to {in_channels} for weight '{name}'"
    if group_size:
        if group_size > in_channels:
            logger.debug(msg + f" as it exceeds the number of input channels {in_channels}.")
            group_size = in_channels
        if in_channels % group_size != 0:
            logger.debug(msg + f" as it does not divide the number of input channels {in_channels}.")
            group_size = in_channels
    return group_size


def is_matmul_nbits_compatible(qconfig: QConfig, name: str = "") -> bool:
    """_common.py:32-62: weight-only, uint4 / uint8, group strategy, group_size -1 or a power of two >= 16."""
    msg = f"Found uncompatibility for MatMulNBits in {name}: "
    if not (qconfig.input_activations is None and qconfig.output_activations is None):
        logger.debug(msg + "It only supports weight-only quantization.")
        return False
    if qconfig.weights.dtype not in {QuantType.QUInt4, QuantType.QUInt8}:
        logger.debug(msg + f"It only supports uint4 and uint8 weight types. Found: {qconfig.weights.dtype}")
        return False
    if qconfig.weights.strategy != QuantizationStrategy.GROUP:
        logger.debug(msg + "It only supports 'group' quantization strategy. Found: " + str(qconfig.weights.strategy))
        return False
    g = qconfig.weights.group_size
    if g != -1 and (g < 16 or (g & (g - 1)) != 0):
        logger.debug(msg + "group_size should be a power of 2 greater than or equal to 16.")
        return False
    return True


def _prepare_for_matmul_nbits(w_q: np.ndarray, w_scale: np.ndarray, w_zero_point: np.ndarray, qconfig: QConfig):
    """_common.py:65-123: (B [N, K/g, g*bits/8] uint8, scales [N, K/g], zero points [N, ceil(K/g / 2)] packed | [N, K/g])."""
    import torch

    from .hip import ops

    in_channels, out_channels = w_q.shape
    bits = qconfig.weights.dtype.bitwidth
    g = qconfig.weights.group_size
    assert in_channels % g == 0
    blocks = in_channels // g
    q_dev = torch.from_numpy(np.ascontiguousarray(w_q).view(np.uint8).reshape(w_q.shape)).cuda()
    blob = ops.pack_matmul_nbits(q_dev, g, bits).cpu().numpy()
    scale = np.asarray(w_scale).reshape(-1, blocks)
    float_zp = qconfig.weights.zp_dtype == scale.dtype
    if bits == 4 and blocks > 1 and not float_zp:
        zp_dev = torch.from_numpy(np.ascontiguousarray(w_zero_point).astype(np.uint8).reshape(-1)).cuda()
        packed = ops.pack_zero_points_u4(zp_dev, out_channels, blocks).cpu().numpy()
    else:
        packed = np.asarray(w_zero_point)
    zp_dtype = qconfig.weights.zp_dtype if float_zp else np.uint8
    return blob, scale, np.reshape(packed, (out_channels, -1)).astype(zp_dtype)
