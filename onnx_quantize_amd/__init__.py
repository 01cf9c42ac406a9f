"""onnx_quantize_amd -- MI355X-native numeric hot path of onnx_quantize behind the reference's API.

Public names mirror ``onnx_quantize/__init__.py`` of the reference for the path in scope
(SURVEY.md section 8): configuration objects, the RTN / GPTQ / HQQ algorithm plugins, the calibration
parameters and ``quantize``.  Importing this package never touches the GPU; the HIP library is
loaded on first use and its absence is an error, never a silent CPU fallback.
"""
from ._logging import set_log_level  # noqa: F401
from .algorithms.gptq import GPTQConfig  # noqa: F401
from .algorithms.hqq import HqqConfig  # noqa: F401
from .algorithms.rtn import RTNConfig  # noqa: F401
from .calibration import CalibrationMethod, CalibrationParams  # noqa: F401
from .config import (  # noqa: F401
    AlgorithmConfig,
    PreProcessingConfig,
    QActivationArgs,
    QConfig,
    QFormat,
    QuantizationStrategy,
    QWeightArgs,
    register_algorithm_config,
    register_preprocessing_config,
)
from .dtypes import QuantType  # noqa: F401
from .preprocessing import AwqConfig, SmoothQuantConfig  # noqa: F401
from .quantize import quantize  # noqa: F401

__version__ = "0.1.0"
