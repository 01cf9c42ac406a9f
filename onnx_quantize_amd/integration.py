"""Glue for running the reference's graph pipeline on top of the HIP numeric path.

The reference reaches its numeric code through one function, ``qrules/_common.py::quantize_weights`` (:126-142, the sole
caller of ``qconfig.weights.algorithm.quantize_weights`` at :133 and of ``_prepare_for_matmul_nbits`` at :137), plus the
calibrator registry (core/_calibration/factory.py:10) and two helpers the AWQ pass binds by name.  Every rule module
imports ``quantize_weights`` BY NAME (qrules/_qdq/matmul_to_qmatmul.py:8, _qdq/gemm_to_qgemm.py:5, _qlinear/*.py:6), so
the replacement is rebound in each of those namespaces: after ``install_into_reference()`` a MatMulNBits-compatible
weight goes upload -> fused RTN/HQQ kernel writing the blob -> packed zero points -> one download (seam.py), and never
through the reference's NumPy packer.
"""
from __future__ import annotations

import importlib
import logging

logger = logging.getLogger(__name__)

# modules of the reference that hold `quantize_weights` in their namespace
_RULE_MODULES = (
    "onnx_quantize.qrules._common",
    "onnx_quantize.qrules._qdq.matmul_to_qmatmul",
    "onnx_quantize.qrules._qdq.gemm_to_qgemm",
    "onnx_quantize.qrules._qlinear.matmul_to_qmatmul",
    "onnx_quantize.qrules._qlinear.gemm_to_qgemm",
)

_installed: dict = {}


def _rebind_seam() -> list[str]:
    from . import seam

    done = []
    for name in _RULE_MODULES:
        try:
            mod = importlib.import_module(name)
        except ImportError:
            continue                                   # the rule modules need onnxscript; the numeric modules do not
        if hasattr(mod, "quantize_weights"):
            mod.quantize_weights = seam.quantize_weights
            done.append(name)
    return done


def _register_extended_configs(ref_cfg, ref_gptq) -> None:
    """``quantize()`` serialises the config into every node (`model_dump`, pre_passes/__init__.py:23) and re-parses it per
    node through the reference's registry (qrules/base.py:57).  The reference's GPTQConfig has no ``mode`` field and
    pydantic drops unknown keys, so a user's ``mode="corrected"`` would silently become parity.  Register a subclass of
    the reference's class that carries the field under the same tag."""
    from typing import Literal

    if getattr(ref_cfg._ALGORITHM_REGISTRY.get("gptq"), "_oq_extended", False):
        return

    class GPTQConfig(ref_gptq.GPTQConfig):             # same tag "gptq", same fields + mode
        _oq_extended = True
        mode: Literal["parity", "corrected"] = "parity"

        def quantize_weights(self, w, qconfig, out=None):      # gptq.py:51-73 with `mode` carried along
            from .seam import weight_arrays

            return weight_arrays(w, qconfig, out, False)

    GPTQConfig.__qualname__ = "GPTQConfig"
    ref_cfg._ALGORITHM_REGISTRY["gptq"] = GPTQConfig


def _extend_providers(ref_base, ref_calibrate=None) -> None:
    """The two ROCm execution providers (calibration.ExecutionProvider) are additions; the reference's enum rejects them
    (base.py:12-32).  A superset enum with the reference's member names, values and aliases takes its place in the two
    modules that look the name up at call time (base.py:83, calibrate.py:340)."""
    from .calibration import ExecutionProvider

    ref_base.ExecutionProvider = ExecutionProvider
    if ref_calibrate is not None:
        ref_calibrate.ExecutionProvider = ExecutionProvider


def install_into_reference() -> dict:
    """Replace the reference's numeric plugins by the HIP ones (requires ``onnx_quantize`` importable).  Returns what was
    rebound (for logs and tests)."""
    import onnx_quantize.core._algorithms.gptq as ref_gptq
    import onnx_quantize.core._algorithms.hqq as ref_hqq
    import onnx_quantize.core._algorithms.rtn as ref_rtn
    import onnx_quantize.core._calibration.base as ref_base
    import onnx_quantize.core._calibration.factory as ref_factory
    import onnx_quantize.core._qconfig as ref_cfg

    from .algorithms.functional import _dequantize_array
    from .algorithms.gptq import _gptq_quantize
    from .algorithms.hqq import _hqq_quantize
    from .algorithms.rtn import _rtn_quantize
    from .calibration import MinMaxCalibrator

    ref_rtn._rtn_quantize = _rtn_quantize          # rtn.py:37-51 resolves the name at call time
    ref_gptq._gptq_quantize = _gptq_quantize       # gptq.py:51-73 likewise
    ref_hqq._hqq_quantize = _hqq_quantize          # hqq.py:80-97 likewise
    ref_factory._CALIBRATORS[ref_factory.CalibrationMethod.MINMAX] = MinMaxCalibrator
    _register_extended_configs(ref_cfg, ref_gptq)
    try:
        import onnx_quantize.core._calibration.calibrate as ref_calibrate
    except ImportError:
        ref_calibrate = None
    _extend_providers(ref_base, ref_calibrate)
    rebound = {"algorithms": ["_rtn_quantize", "_gptq_quantize", "_hqq_quantize"], "calibrator": "minmax",
               "quantize_weights": _rebind_seam(), "awq": False, "awq_pass": [], "smooth_quant_pass": [], "calibrate": [],
               "cleanup": False}
    # the AWQ pass binds the two helpers by name at import time (pre_passes/awq.py:10-11): rebind them in its namespace --
    # and, since round 4, its two search methods themselves (one device call each instead of a 20- / 10-iteration host loop
    # with a PCIe round trip per candidate), SmoothQuant's node method and the calibration walks (reference_passes.py)
    from . import reference_passes

    try:
        import onnx_quantize.pre_passes.awq as ref_awq

        ref_awq._rtn_quantize = _rtn_quantize
        ref_awq._dequantize_array = _dequantize_array
        reference_passes.install_awq(ref_awq)
        rebound["awq"] = True
        rebound["awq_pass"] = ["AwqPass._apply_awq", "AwqPass._apply_awq_clip"]
    except ImportError:
        pass
    try:
        import onnx_quantize.pre_passes.smooth_quant as ref_sq

        reference_passes.install_smooth_quant(ref_sq)
        rebound["smooth_quant_pass"] = ["SmoothQuantPass._smooth_quant_node"]
    except ImportError:
        pass
    if ref_calibrate is not None:
        import sys

        # pre_passes/__init__.py:8 imported `calibrate_model` by name (that package needs onnxscript: only if it is loaded)
        reference_passes.install_calibrate(ref_calibrate, also=[m for m in (sys.modules.get("onnx_quantize.pre_passes"),) if m is not None])
        rebound["calibrate"] = ["_set_qparams", "_set_qparams_gptq", "calibrate_model"]
    # the rewrite is over when quantize.py:68 asks for the emitted functions (`get_qfunctions`, resolved by name in that
    # module at call time, so this also covers callers that bound `quantize` before the install): drop the cached
    # Hessians / factors of the GPTQ nodes there, on the drop-in path too
    try:
        ref_quantize = importlib.import_module("onnx_quantize.quantize")   # the module (the package attribute is the function)

        if not getattr(ref_quantize.get_qfunctions, "_oq_cleanup", False):
            original = ref_quantize.get_qfunctions

            def get_qfunctions(*args, **kwargs):
                from .seam import clear_shared_inputs

                clear_shared_inputs()
                return original(*args, **kwargs)

            get_qfunctions._oq_cleanup = True
            ref_quantize.get_qfunctions = get_qfunctions
        rebound["cleanup"] = True
    except ImportError:
        pass
    _installed.update(rebound)
    return rebound


def reference_qconfig(qconfig):
    """This package's QConfig as the reference's, field for field.  Extension fields the reference would drop are either
    carried by the registered subclasses (GPTQ ``mode``) or rejected loudly."""
    import onnx_quantize as ref

    install_into_reference()
    dumped = qconfig.model_dump()
    ref_qconfig = ref.QConfig(**dumped)

    def lost(ours, theirs, path):
        if isinstance(ours, dict) and hasattr(theirs, "model_dump"):
            theirs = theirs.model_dump()
        if isinstance(ours, dict) and isinstance(theirs, dict):
            for k, v in ours.items():
                if k not in theirs:
                    return f"{path}.{k}" if v is not None else None
                hit = lost(v, theirs[k], f"{path}.{k}")
                if hit:
                    return hit
        return None

    missing = lost(dumped, ref_qconfig.model_dump(), "qconfig")
    if missing:
        raise ValueError(f"{missing} is an extension of onnx_quantize_amd that the installed reference would silently drop")
    return ref_qconfig


def quantize_with_reference_pipeline(model, qconfig):
    try:
        import onnx_quantize as ref
    except ImportError as e:
        raise ImportError(
            "graph rewriting is delegated to the reference package `onnx_quantize`, which is not installed; "
            "only the numeric path (onnx_quantize_amd.algorithms / .hip.ops / .seam) is available") from e
    ref_qconfig = reference_qconfig(qconfig)
    try:
        return ref.quantize(model, ref_qconfig)
    finally:
        from . import seam

        seam.clear_shared_inputs()
