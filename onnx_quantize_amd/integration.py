"""Glue for running the reference's graph pipeline on top of the HIP numeric path.

The reference calls its numeric code through one seam, ``qconfig.weights.algorithm.quantize_weights``
(qrules/_common.py:133), plus the calibrator registry (core/_calibration/factory.py:10).  Swapping the
registered classes is therefore the whole integration.
"""
from __future__ import annotations


def install_into_reference() -> None:
    """Replace the reference's numeric plugins by the HIP ones (requires ``onnx_quantize`` importable)."""
    import onnx_quantize.core._calibration.factory as ref_factory
    import onnx_quantize.core._qconfig as ref_cfg

    from .algorithms.gptq import _gptq_quantize
    from .algorithms.hqq import _hqq_quantize
    from .algorithms.rtn import _rtn_quantize
    from .calibration import MinMaxCalibrator

    import onnx_quantize.core._algorithms.gptq as ref_gptq
    import onnx_quantize.core._algorithms.hqq as ref_hqq
    import onnx_quantize.core._algorithms.rtn as ref_rtn

    ref_rtn._rtn_quantize = _rtn_quantize          # rtn.py:37-51 resolves the name at call time
    ref_gptq._gptq_quantize = _gptq_quantize       # gptq.py:51-73 likewise
    ref_hqq._hqq_quantize = _hqq_quantize          # hqq.py:80-97 likewise
    ref_factory._CALIBRATORS[ref_factory.CalibrationMethod.MINMAX] = MinMaxCalibrator
    # the AWQ pass binds the two helpers by name at import time (pre_passes/awq.py:10-11): rebind them in its namespace
    try:
        import onnx_quantize.pre_passes.awq as ref_awq

        from .algorithms.functional import _dequantize_array

        ref_awq._rtn_quantize = _rtn_quantize
        ref_awq._dequantize_array = _dequantize_array
    except ImportError:
        pass
    del ref_cfg


def quantize_with_reference_pipeline(model, qconfig):
    try:
        import onnx_quantize as ref
    except ImportError as e:
        raise ImportError(
            "graph rewriting is delegated to the reference package `onnx_quantize`, which is not installed; "
            "only the numeric path (onnx_quantize_amd.algorithms / .hip.ops) is available") from e
    install_into_reference()
    ref_qconfig = ref.QConfig(**qconfig.model_dump())
    return ref.quantize(model, ref_qconfig)
