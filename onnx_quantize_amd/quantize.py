"""Top-level ``quantize(model, qconfig)`` (reference: quantize.py:28-80).

Two routes, chosen by what the caller hands over:

* ONNX **bytes**, a **path** or a model parsed by `onnx_proto.parse_model`: this package's own writer (`model_quantize`,
  SURVEY.md 8f rows N4 / N1) -- no `onnx`, `onnx_ir` or `onnxscript` needed.  Bytes in -> bytes out, path in -> parsed model
  out (save it with `onnx_proto.save_model`), parsed model in -> parsed model out.
* an ``onnx.ModelProto`` / ``onnx_ir.Model`` (those packages installed): with the reference package importable its own
  pipeline runs with the two numeric plugins swapped for the HIP ones (``RTNConfig`` / ``GPTQConfig`` of this package
  register under the same tags; integration.py); without it a ModelProto goes through this package's writer by way of its
  serialised bytes and comes back as a ModelProto.

Anything else is the reference's TypeError.  Nothing here computes on the CPU: the numbers come from the HIP library, and
its absence is an error.
"""
from __future__ import annotations

import logging
import os

from .config import QConfig

__all__ = ["quantize"]

logger = logging.getLogger("onnx_quantize")


def quantize(model, qconfig: QConfig):
    """Same signature and error behaviour as the reference: TypeError for anything that is not a model; the model is
    returned unchanged when ``qconfig`` selects nothing to quantize."""
    from .onnx_proto import Message, serialize

    own_input = isinstance(model, (bytes, bytearray, memoryview, str, os.PathLike)) or \
        (isinstance(model, Message) and model._type == "ModelProto")
    if own_input:
        from .model_quantize import quantize_model

        nothing = qconfig.weights is None and qconfig.input_activations is None and qconfig.output_activations is None
        if nothing and not isinstance(model, (str, os.PathLike)):
            logger.info("Nothing to quantize: returning the model unchanged.")
            return model
        out = quantize_model(model, qconfig)
        return serialize(out) if isinstance(model, (bytes, bytearray, memoryview)) else out

    try:
        import onnx
    except ImportError as e:
        if not type(model).__module__.split(".")[0].startswith("onnx"):      # quantize.py:38-41: not a model of any kind
            raise TypeError(f"model must be ONNX bytes, a path, a parsed ModelProto, an onnx.ModelProto or an onnx_ir.Model, got {type(model)}") from None
        raise ImportError(
            "quantize() takes ONNX bytes, a path or a model parsed by onnx_quantize_amd.onnx_proto.parse_model; an object of "
            f"type {type(model).__name__} needs the `onnx` / `onnx_ir` packages, which are not installed here") from e
    try:
        import onnx_ir as ir
        model_types = (onnx.ModelProto, ir.Model)
    except ImportError:
        ir, model_types = None, (onnx.ModelProto,)
    if not isinstance(model, model_types):
        raise TypeError(f"model must be an instance of onnx.ModelProto or onnx_ir.Model, got {type(model)}")
    if qconfig.weights is None and qconfig.input_activations is None and qconfig.output_activations is None:
        logger.info("Nothing to quantize: returning the model unchanged.")
        return model
    try:
        import onnx_quantize  # noqa: F401  (the reference: its own graph pipeline with the HIP plugins)
        have_reference = ir is not None
    except ImportError:
        have_reference = False
    if have_reference:
        from .integration import quantize_with_reference_pipeline

        return quantize_with_reference_pipeline(model, qconfig)
    if not isinstance(model, onnx.ModelProto):
        raise ImportError("an onnx_ir.Model needs the reference package `onnx_quantize` for its graph pipeline; pass ONNX bytes "
                          "or a path to use this package's own writer")
    from .model_quantize import quantize_model

    return onnx.ModelProto.FromString(serialize(quantize_model(model.SerializeToString(), qconfig)))
