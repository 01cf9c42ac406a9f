"""Pre-processing configurations (reference: pre_passes/awq.py:24-37, pre_passes/smooth_quant.py:20-35) and the
device-resident numeric cores of their passes -- SURVEY.md 8f, row N2.

The passes themselves are graph surgery (Mul insertion, initializer replacement): ``build_pass`` hands over to the
reference's pass classes when the ONNX stack is importable (and raises otherwise); on ONNX bytes / files this package's
own writer does the same surgery (`model_quantize._preprocess`).  What runs on the GPU in both is
the arithmetic inside them -- the AWQ 20-point scale grid and 10-point clip search, the SmoothQuant scale -- exposed
as ``awq_scale_search`` / ``awq_clip_search`` / ``smooth_quant_scale`` (NumPy in / out, device resident in between).
"""
from __future__ import annotations

from typing import Literal

import numpy as np

from .config import PreProcessingConfig, QuantizationStrategy, register_preprocessing_config
from .dtypes import QuantType

__all__ = ["AwqConfig", "SmoothQuantConfig", "awq_scale_search", "awq_clip_search", "smooth_quant_scale"]


def _reference_pass(module: str, cls: str, **kwargs):
    try:
        import importlib

        from .integration import install_into_reference

        install_into_reference()
        return getattr(importlib.import_module(f"onnx_quantize.pre_passes.{module}"), cls)(**kwargs)
    except ImportError as e:
        raise ImportError(
            f"{cls} rewrites an ONNX graph and is delegated to the reference package `onnx_quantize`, which is not "
            "installed; the numeric cores are available as onnx_quantize_amd.preprocessing.awq_scale_search / "
            "awq_clip_search / smooth_quant_scale") from e


@register_preprocessing_config
class AwqConfig(PreProcessingConfig):
    """awq.py:24-37.  Args: clip_search -- also search the clip ratio (default False)."""

    preprocessing_type: Literal["awq"] = "awq"
    clip_search: bool = False

    def build_pass(self, qconfig):
        return _reference_pass("awq", "AwqPass", clip_search=self.clip_search, target_op_types=qconfig.target_op_types)


@register_preprocessing_config
class SmoothQuantConfig(PreProcessingConfig):
    """smooth_quant.py:20-35.  Args: alpha -- how much of the activation range moves into the weights (default 0.5)."""

    preprocessing_type: Literal["smooth_quant"] = "smooth_quant"
    alpha: float = 0.5

    def build_pass(self, qconfig):
        return _reference_pass("smooth_quant", "SmoothQuantPass", alpha=self.alpha, target_op_types=qconfig.target_op_types)


def _dev(a):
    import torch

    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).cuda()


def _key(strategy) -> str:
    return strategy.value if isinstance(strategy, QuantizationStrategy) else str(strategy)


def awq_scale_search(inputs: np.ndarray, weights: np.ndarray, quant_type: QuantType, strategy, group_size, is_symmetric=False,
                     reduce_range=False):
    """awq.py:114-184 without the graph edits: (best_scale [K] fp32, losses [20])."""
    from .hip import ops

    s, losses = ops.awq_scale_search(_dev(inputs), _dev(weights), quant_type.key, _key(strategy), group_size, bool(is_symmetric),
                                     bool(reduce_range))
    return s.cpu().numpy(), losses


def awq_clip_search(inputs: np.ndarray, weights: np.ndarray, quant_type: QuantType, strategy, group_size, is_symmetric=False,
                    reduce_range=False):
    """awq.py:207-259: (best clip_ratio, losses [10])."""
    from .hip import ops

    return ops.awq_clip_search(_dev(inputs), _dev(weights), quant_type.key, _key(strategy), group_size, bool(is_symmetric),
                               bool(reduce_range))


def smooth_quant_scale(inputs: np.ndarray, weights: np.ndarray, alpha: float = 0.5) -> np.ndarray:
    """smooth_quant.py:104-113: the per-input-channel smoothing scale [K]."""
    from .hip import ops

    return ops.smooth_quant_scale(_dev(inputs), _dev(weights), float(alpha)).cpu().numpy()
