"""GPTQ weight quantization on the GPU (reference: core/_algorithms/gptq.py).

The default ``mode="parity"`` reproduces the reference exactly as written -- including the fact that its
error-feedback terms index the zero half of the upper-triangular inverse factor (SURVEY.md finding 1), so
the emitted integers equal RTN with per-group parameters.  ``mode="corrected"`` applies the update GPTQ
intends; it is opt-in, documented as NOT bit-compatible with the reference, and returns the parameters the
integers were produced with.
"""
from __future__ import annotations

import logging
from typing import TYPE_CHECKING, ClassVar, Literal

import numpy as np

from ..config import AlgorithmConfig, QuantizationStrategy, register_algorithm_config
from ..dtypes import QuantType

if TYPE_CHECKING:  # pragma: no cover
    import onnx_ir as ir

    from ..config import QConfig

__all__ = ["GPTQConfig", "_gptq_quantize"]

logger = logging.getLogger(__name__)

_FALLBACK_WARNING = (
    "Failed to invert hessian due to numerical instability. Consider increasing percdamp, increasing the "
    "number of calibration samples, or shuffling the calibration dataset. Falling back to round-to-nearest "
    "for this module.")


@register_algorithm_config
class GPTQConfig(AlgorithmConfig):
    """gptq.py:34-73: block_size=128, percdamp=0.01, actorder=False; needs calibration activations.
    ``mode`` is an extension ("parity" = the reference as written, "corrected" = intended update)."""

    requires_calibration: ClassVar[bool] = True

    algorithm_type: Literal["gptq"] = "gptq"
    block_size: int = 128
    percdamp: float = 0.01
    actorder: bool = False
    mode: Literal["parity", "corrected"] = "parity"

    def quantize_weights(self, w: "ir.Value", qconfig: "QConfig", out: "ir.Value | None" = None):
        assert out is not None, "Output value is required for GPTQ quantization."
        node = out.producer()
        assert "input" in node.meta, "GPTQ requires calibration data in node meta."
        a = qconfig.weights
        return _gptq_quantize(w.const_value.numpy(), node.meta["input"], quant_type=a.dtype, strategy=a.strategy,
                              is_symmetric=a.symmetric, reduce_range=a.reduce_range, clip_ratio=a.clip_ratio,
                              block_size=self.block_size, percdamp=self.percdamp, group_size=a.group_size,
                              actorder=self.actorder, mse=a.mse, scale_dtype=a.scale_dtype, zp_dtype=a.zp_dtype,
                              mode=self.mode)


def _accumulate_hessian(inp, H, num_samples):
    """gptq.py:246-260 with ``H`` a torch tensor in HBM (updated in place); ``inp`` NumPy or torch."""
    import torch

    from ..hip import ops

    x = inp if isinstance(inp, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(inp, dtype=np.float32))
    if not x.is_cuda:
        x = x.cuda()
    return H, ops.hessian_accumulate(x.to(torch.float32), H, num_samples)


def _gptq_quantize(weights, inputs, quant_type=QuantType.QInt8, strategy=QuantizationStrategy.CHANNEL, group_size=32,
                   is_symmetric=False, reduce_range=False, clip_ratio=1.0, block_size=128, percdamp=0.01,
                   actorder=False, mse=False, scale_dtype=np.float32, zp_dtype=np.int8, *, mode="parity",
                   batch_rows=None):
    """gptq.py:263-324.  ``inputs`` [num_samples, ..., in_features]; a list / iterator of such batches is
    also accepted and streamed into the Hessian (the reference concatenates every batch in host memory).
    """
    import torch

    from ..hip import ops

    from ..staging import upload

    from ..reference_passes import StreamedGptqInput

    w = upload(np.asarray(weights))
    k = w.shape[0]
    if isinstance(inputs, StreamedGptqInput):                              # H accumulated by the calibration walk (calibrate.py:288-307 rebound)
        if tuple(inputs.h.shape) != (k, k):
            raise ValueError(f"streamed Hessian of '{inputs.name}' is {tuple(inputs.h.shape)}, the weight has {k} input channels")
        h = inputs.h
    else:
        h = torch.zeros((k, k), dtype=torch.float32, device=w.device)      # gptq.py:304
        n = 0
        batches = inputs if isinstance(inputs, (list, tuple)) or hasattr(inputs, "__next__") else [inputs]
        for x in batches:
            h, n = _accumulate_hessian(x, h, n)                            # :305
    q, s, z, info = ops.gptq_quantize(w, h, quant_type.key, strategy.value, group_size, bool(is_symmetric),
                                      bool(reduce_range), float(clip_ratio), int(block_size), float(percdamp),
                                      bool(actorder), bool(mse), mode=mode)
    if int(info.item()) != 0:                                              # :143-150
        logger.warning(_FALLBACK_WARNING)
    from ..staging import download

    q_np = download(q).astype(quant_type.np_dtype, copy=False)
    scale = s.cpu().numpy().astype(np.float32, copy=False)                 # :238
    zp = z.cpu().numpy().astype(q_np.dtype, copy=False)                    # :239
    return q_np, scale, zp
