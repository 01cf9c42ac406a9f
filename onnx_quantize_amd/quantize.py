"""Top-level ``quantize(model, qconfig)`` (reference: quantize.py:28-80).

Only the numeric hot path is re-implemented in this repository; graph surgery --
pre-passes, rewrite rules and the emitted ``quant``-domain functions -- is the reference's own Python and
needs ``onnx`` / ``onnx_ir`` / ``onnxscript``.  When those packages and the reference are importable,
``quantize`` delegates to the reference pipeline with its two numeric plugins swapped for the HIP ones
(``RTNConfig`` / ``GPTQConfig`` of this package register under the same tags).  Without them it fails
loudly instead of pretending.
"""
from __future__ import annotations

import logging

from .config import QConfig

__all__ = ["quantize"]

logger = logging.getLogger("onnx_quantize")


def quantize(model, qconfig: QConfig):
    """Same signature and error behaviour as the reference: TypeError for anything that is not an
    ``onnx.ModelProto`` / ``onnx_ir.Model``; the model is returned unchanged when ``qconfig`` selects
    nothing to quantize."""
    try:
        import onnx
        import onnx_ir as ir
    except ImportError as e:
        raise ImportError(
            "quantize() rewrites an ONNX graph and needs the `onnx`, `onnx_ir` and `onnxscript` packages, "
            "which are not installed here.  The numeric path is usable on its own: see "
            "onnx_quantize_amd.algorithms._rtn_quantize / _gptq_quantize and INTEGRATION.md.") from e
    if not isinstance(model, (onnx.ModelProto, ir.Model)):
        raise TypeError(f"model must be an instance of onnx.ModelProto or onnx_ir.Model, got {type(model)}")
    if qconfig.weights is None and qconfig.input_activations is None and qconfig.output_activations is None:
        logger.info("Nothing to quantize: returning the model unchanged.")
        return model
    from .integration import quantize_with_reference_pipeline

    return quantize_with_reference_pipeline(model, qconfig)
