"""Half-quadratic quantization (HQQ) on the GPU (reference: core/_algorithms/hqq.py) -- SURVEY.md 8f, row N2."""
from __future__ import annotations

from typing import TYPE_CHECKING, Literal

import numpy as np

from ..config import AlgorithmConfig, QuantizationStrategy, register_algorithm_config
from ..dtypes import QuantType

if TYPE_CHECKING:  # pragma: no cover
    import onnx_ir as ir

    from ..config import QConfig, QWeightArgs

__all__ = ["HqqConfig", "_hqq_quantize"]


@register_algorithm_config
class HqqConfig(AlgorithmConfig):
    """hqq.py:27-97: same fields, defaults, constraints and error messages as the reference.

    Args:
        lp_norm: the Lp norm of the half-quadratic solver (default 0.7).
        beta: shrinkage parameter (default 10.0), multiplied by ``kappa`` (default 1.01) every round.
        iters: number of rounds (default 20).
        early_stop: stop at the first round that does not lower the mean error (default True).
    """

    algorithm_type: Literal["hqq"] = "hqq"
    lp_norm: float = 0.7
    beta: float = 1e1
    kappa: float = 1.01
    iters: int = 20
    early_stop: bool = True

    @staticmethod
    def _check_hqq_constraints(dtype: QuantType, symmetric: bool, strategy: QuantizationStrategy, group_size: int) -> None:
        if dtype != QuantType.QUInt4:
            raise ValueError(f"HQQ only supports uint4 weight type. Found: {np.dtype}")   # message as in hqq.py:53
        if symmetric:
            raise ValueError("HQQ only supports asymmetric quantization.")
        if strategy != QuantizationStrategy.GROUP:
            raise ValueError(f"HQQ only supports 'group' quantization strategy. Found: {strategy}")
        if group_size != -1 and (group_size < 16 or (group_size & (group_size - 1)) != 0):
            raise ValueError(
                f"HQQ requires group_size to be greater than 16 and a power of 2. Found: {group_size}")

    def validate_weight_args(self, weight_args: "QWeightArgs") -> None:
        self._check_hqq_constraints(weight_args.dtype, weight_args.symmetric, weight_args.strategy, weight_args.group_size)
        weight_args.zp_dtype = weight_args.scale_dtype            # hqq.py:77-78: zero points are floats

    def quantize_weights(self, w: "ir.Value", qconfig: "QConfig", out: "ir.Value | None" = None):
        a = qconfig.weights
        return _hqq_quantize(w.const_value.numpy(), quant_type=a.dtype, group_size=a.group_size, reduce_range=a.reduce_range,
                             clip_ratio=a.clip_ratio, mse=a.mse, scale_dtype=a.scale_dtype, zp_dtype=a.zp_dtype,
                             lp_norm=self.lp_norm, beta=self.beta, kappa=self.kappa, iters=self.iters,
                             early_stop=self.early_stop)


def _hqq_quantize(w_f: np.ndarray, quant_type: QuantType, group_size: int, reduce_range: bool = False, clip_ratio: float = 1.0,
                  mse: bool = False, scale_dtype: np.dtype = np.float32, zp_dtype: np.dtype = np.float32, lp_norm: float = 0.7,
                  beta: float = 1e1, kappa: float = 1.01, iters: int = 20, early_stop: bool = True):
    """hqq.py:147-213 through oq_hqq_optimize_f32: (q [K, N], scale [N*K/g, 1], zero_point [N*K/g, 1] float)."""
    import torch

    from ..hip import ops

    assert zp_dtype == scale_dtype                                   # hqq.py:175
    if quant_type != QuantType.QUInt4:
        raise ValueError("the GPU HQQ path implements the reference's only legal configuration: uint4")
    w = np.asarray(w_f)
    from ..staging import upload

    wd = upload(w)
    q, s, z, _ = ops.hqq_quantize(wd, -1 if group_size is None else group_size, bool(reduce_range), float(clip_ratio), bool(mse),
                                  float(lp_norm), float(beta), float(kappa), int(iters), bool(early_stop))
    from ..staging import download

    q_np = download(q).astype(quant_type.np_dtype, copy=False)
    return q_np, s.cpu().numpy().astype(scale_dtype, copy=False), z.cpu().numpy().astype(zp_dtype, copy=False)
