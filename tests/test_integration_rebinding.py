"""INTEGRATION.md section 1 against the reference's real modules: after `install_into_reference()` the reference's plugin
classes reach this package's functions.  Needs the reference checkout (build container only: skipped elsewhere) and no
GPU -- nothing is computed, only the bindings are inspected."""
import importlib.util
import os
import sys

import pytest

REF = "/root/reference/src/onnx_quantize"
pytestmark = pytest.mark.skipif(not os.path.isdir(REF), reason="the reference checkout is only present in the build container")


@pytest.fixture(scope="module")
def reference_modules():
    here = os.path.dirname(os.path.abspath(__file__))
    spec = importlib.util.spec_from_file_location("make_golden", os.path.join(here, "golden", "make_golden.py"))
    mg = importlib.util.module_from_spec(spec)
    saved = {k: sys.modules.get(k) for k in ("onnx_ir", "onnx_quantize", "ml_dtypes")}
    spec.loader.exec_module(mg)              # installs the stand-ins of tests/golden/PROVENANCE.json and imports the numeric modules
    mg._load_passes()
    mg.originals = dict(rtn=mg.R.rtn._rtn_quantize, gptq=mg.R.gptq._gptq_quantize, hqq=mg.R.hqq._hqq_quantize)   # before any swap
    yield mg
    for k in [m for m in sys.modules if m == "onnx_quantize" or m.startswith("onnx_quantize.")] + ["onnx_ir", "ml_dtypes"]:
        sys.modules.pop(k, None)
    for k, v in saved.items():
        if v is not None:
            sys.modules[k] = v


def test_install_into_reference_rebinds_what_the_plugins_look_up(reference_modules):
    import onnx_quantize.core._algorithms.gptq as ref_gptq
    import onnx_quantize.core._algorithms.hqq as ref_hqq
    import onnx_quantize.core._algorithms.rtn as ref_rtn
    import onnx_quantize.core._calibration.factory as ref_factory
    import onnx_quantize.pre_passes.awq as ref_awq

    from onnx_quantize_amd import integration
    from onnx_quantize_amd.algorithms import _gptq_quantize, _hqq_quantize, _rtn_quantize
    from onnx_quantize_amd.algorithms.functional import _dequantize_array
    from onnx_quantize_amd.calibration import MinMaxCalibrator

    # the plugin methods resolve the function through the module globals at CALL time: that is what makes the swap work
    assert "_rtn_quantize" in ref_rtn.RTNConfig.quantize_weights.__code__.co_names
    assert "_gptq_quantize" in ref_gptq.GPTQConfig.quantize_weights.__code__.co_names
    assert "_hqq_quantize" in ref_hqq.HqqConfig.quantize_weights.__code__.co_names
    before = ref_rtn._rtn_quantize
    integration.install_into_reference()
    assert before is not _rtn_quantize
    assert ref_rtn._rtn_quantize is _rtn_quantize and ref_rtn.RTNConfig.quantize_weights.__globals__["_rtn_quantize"] is _rtn_quantize
    assert ref_gptq.GPTQConfig.quantize_weights.__globals__["_gptq_quantize"] is _gptq_quantize
    assert ref_hqq.HqqConfig.quantize_weights.__globals__["_hqq_quantize"] is _hqq_quantize
    assert ref_factory._CALIBRATORS[ref_factory.CalibrationMethod.MINMAX] is MinMaxCalibrator
    cal = ref_factory.get_calibrator(ref_factory.CalibrationMethod.MINMAX, momentum=0.5)       # no GPU needed to construct
    assert isinstance(cal, MinMaxCalibrator) and cal.momentum == 0.5 and cal.data == {}
    with pytest.raises(TypeError, match="Invalid arguments for MinMaxCalibrator"):
        ref_factory.get_calibrator(ref_factory.CalibrationMethod.MINMAX, nope=1)
    # the AWQ pass bound its two helpers at import time: rebound in its namespace
    assert ref_awq._rtn_quantize is _rtn_quantize and ref_awq._dequantize_array is _dequantize_array
    # ... and, since round 4, its two search methods themselves are this package's (tests/test_reference_passes.py)
    assert getattr(ref_awq.AwqPass._apply_awq, "_oq_rebound", False) and getattr(ref_awq.AwqPass._apply_awq_clip, "_oq_rebound", False)


def test_signatures_of_the_swapped_functions_match(reference_modules):
    """Same parameter names, order and defaults as the functions they replace (rtn.py:54-65, gptq.py:263-279, hqq.py:147-160,
    utils.py): the reference's call sites pass keywords."""
    import inspect

    import onnx_quantize.core._algorithms.utils as ref_utils
    from onnx_quantize_amd.algorithms import _gptq_quantize, _hqq_quantize, _rtn_quantize
    from onnx_quantize_amd.algorithms import functional as F

    import enum

    import numpy as np

    def norm(v):
        if v is inspect._empty:
            return "<required>"
        if isinstance(v, enum.Enum):
            return f"{type(v).__name__}.{v.name}"          # the two packages have their own (equal) enum classes
        if isinstance(v, (np.dtype, type)):
            return np.dtype(v).name
        return v

    def params(fn, drop=()):
        return [(p.name, norm(p.default)) for p in inspect.signature(fn).parameters.values() if p.name not in drop]

    orig = reference_modules.originals
    assert orig["rtn"].__module__.startswith("onnx_quantize.") and orig["gptq"].__module__.startswith("onnx_quantize.")
    assert params(_rtn_quantize) == params(orig["rtn"])
    # `mode` (parity / corrected) and `batch_rows` (streaming granularity) are this package's trailing, defaulted extensions
    assert params(_gptq_quantize, drop=("mode", "batch_rows")) == params(orig["gptq"])
    assert params(_hqq_quantize) == params(orig["hqq"])
    for name in ("_compute_qparams", "_quantize_array_from_qparams", "_dequantize_array", "_fake_quantize_array", "_compute_min_max",
                 "_compute_qparams_from_array", "_preprocess_array", "_post_process_array"):
        assert [p for p, _ in params(getattr(F, name))] == [p for p, _ in params(getattr(ref_utils, name))], name


def test_install_rebinds_the_seam_function_and_keeps_extensions(reference_modules):
    """`qrules/_common.py::quantize_weights` (the function every rule module imports by name) is replaced by the device
    resident seam; the GPTQ `mode` extension survives the dump -> registry round trip of `quantize()`
    (pre_passes/__init__.py:23, qrules/base.py:57); the ROCm execution providers are accepted by the reference's
    CalibrationParams after the install."""
    import inspect

    import onnx_quantize.core._calibration.base as ref_base
    import onnx_quantize.core._qconfig as ref_cfg

    from onnx_quantize_amd import integration, seam

    common = reference_modules._load_common()
    original = common.quantize_weights
    assert original.__module__ == "onnx_quantize.qrules._common"
    rebound = integration.install_into_reference()
    assert "onnx_quantize.qrules._common" in rebound["quantize_weights"]
    assert common.quantize_weights is seam.quantize_weights
    assert list(inspect.signature(seam.quantize_weights).parameters) == list(inspect.signature(original).parameters)
    assert [p.default for p in inspect.signature(seam.quantize_weights).parameters.values()] == \
           [p.default for p in inspect.signature(original).parameters.values()]
    # every rule module the reference ships imports that name: the list in integration.py is complete
    import os
    import re
    users = []
    for root, _, files in os.walk(os.path.join(REF, "qrules")):
        for f in files:
            if f.endswith(".py") and re.search(r"import .*\bquantize_weights\b|^def quantize_weights", open(os.path.join(root, f)).read(), re.M):
                rel = os.path.relpath(os.path.join(root, f), os.path.dirname(REF))[:-3].replace(os.sep, ".")
                users.append(rel)
    assert sorted(users) == sorted(integration._RULE_MODULES)

    # GPTQ `mode` through model_dump() -> QConfig(**dict) as quantize() does
    from onnx_quantize_amd import GPTQConfig, QConfig, QuantType, QWeightArgs
    ours = QConfig(weights=QWeightArgs(dtype=QuantType.QInt4, group_size=32, algorithm=GPTQConfig(mode="corrected", block_size=64)))
    theirs = ref_cfg.QConfig(**ours.model_dump())
    assert theirs.weights.algorithm.mode == "corrected" and theirs.weights.algorithm.block_size == 64
    again = ref_cfg._resolve_algorithm_config(theirs.model_dump()["weights"]["algorithm"])      # the per-node re-parse (_qconfig.py:131-149)
    assert again.mode == "corrected" and again.block_size == 64 and isinstance(again, reference_modules.R.gptq.GPTQConfig)

    # ROCm providers
    from onnx_quantize_amd.calibration import ExecutionProvider
    p = ref_base.CalibrationParams(provider="rocm")          # held as the provider string; calibrate.py:340 re-parses it by name
    assert getattr(p.provider, "value", p.provider) == "ROCMExecutionProvider"
    assert ref_base.ExecutionProvider(p.model_dump()["provider"]) is ExecutionProvider.ROCM
    c = ref_base.CalibrationParams(provider="cpu").provider
    assert getattr(c, "value", c) == "CPUExecutionProvider"
    with pytest.raises(ValueError, match="Invalid execution provider"):
        ref_base.CalibrationParams(provider="tpu")
