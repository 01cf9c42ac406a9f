"""Quantization configuration objects (reference: core/_qconfig.py) -- the API half of the drop-in
boundary.  Same class names, fields, defaults, coercions and exception types as
the reference; the validation rules are re-stated, not copied, and are exercised by
tests/test_config_api.py against the reference's own expectations (test/core/test_qconfig.py).
"""
from __future__ import annotations

import logging
from collections.abc import Sequence
from enum import Enum
from typing import TYPE_CHECKING, Any, ClassVar

import numpy as np
from pydantic import BaseModel, ConfigDict, Field, SerializeAsAny, field_validator, model_validator

from .calibration import CalibrationParams
from .dtypes import QuantType

if TYPE_CHECKING:  # pragma: no cover
    import onnx_ir as ir

logger = logging.getLogger(__name__)

_SUPPORTED_OP_TYPES = ("MatMul", "Gemm")
_FOUR_BIT = (QuantType.QInt4, QuantType.QUInt4)
_EIGHT_BIT = (QuantType.QInt8, QuantType.QUInt8)


class QuantizationStrategy(str, Enum):
    """Granularity of one (scale, zero point) pair."""

    TENSOR = "tensor"
    CHANNEL = "channel"
    GROUP = "group"


class QFormat(str, Enum):
    """Representation of the quantized operators in the emitted graph."""

    QDQ = "qdq"
    QLINEAR = "qlinear"


# --------------------------------------------------------------------------- plugin seams
class AlgorithmConfig(BaseModel):
    """Base class of weight-quantization algorithms (reference: _qconfig.py:46-72).

    A subclass declares a ``Literal`` field ``algorithm_type`` (its registry tag), is decorated with
    :func:`register_algorithm_config`, may set ``requires_calibration`` and implements
    :meth:`quantize_weights` returning ``(q_weight, scale, zero_point)`` as NumPy arrays.
    """

    requires_calibration: ClassVar[bool] = False

    def validate_weight_args(self, weight_args: "QWeightArgs") -> None:
        """Hook for algorithm-specific constraints on the enclosing ``QWeightArgs``."""

    def quantize_weights(self, w: "ir.Value", qconfig: "QConfig", out: "ir.Value | None" = None
                         ) -> tuple[np.ndarray, np.ndarray, np.ndarray]:
        raise NotImplementedError(f"{type(self).__name__} must implement quantize_weights().")


class PreProcessingConfig(BaseModel):
    """Base class of pre-processing passes (reference: _qconfig.py:75-97)."""

    requires_calibration: ClassVar[bool] = True
    requires_post_calibration: ClassVar[bool] = True

    def build_pass(self, qconfig: "QConfig"):
        raise NotImplementedError(f"{type(self).__name__} must implement build_pass().")


_ALGORITHM_REGISTRY: dict[str, type[AlgorithmConfig]] = {}
_PREPROCESSING_REGISTRY: dict[str, type[PreProcessingConfig]] = {}


def _register(cls, registry: dict, tag_field: str):
    field = cls.model_fields.get(tag_field)
    if field is None:
        raise TypeError(f"{cls.__name__} must declare an '{tag_field}' field to be registered.")
    registry[field.default] = cls
    return cls


def register_algorithm_config(cls: type[AlgorithmConfig]) -> type[AlgorithmConfig]:
    """Class decorator: register under the default of the ``algorithm_type`` field."""
    return _register(cls, _ALGORITHM_REGISTRY, "algorithm_type")


def register_preprocessing_config(cls: type[PreProcessingConfig]) -> type[PreProcessingConfig]:
    """Class decorator: register under the default of the ``preprocessing_type`` field."""
    return _register(cls, _PREPROCESSING_REGISTRY, "preprocessing_type")


def _default_algorithm_config() -> AlgorithmConfig:
    from .algorithms.rtn import RTNConfig  # late import: the algorithm modules import this one

    return RTNConfig()


def _from_registry(value: Any, registry: dict, tag_field: str, base: type):
    """Instances pass through; mappings are re-hydrated through the registry (used when a QConfig is
    rebuilt from the ``model_dump`` stored on a graph node)."""
    if isinstance(value, base) or not isinstance(value, dict):
        return value
    tag = value.get(tag_field)
    if tag not in registry:
        raise ValueError(f"Unknown {tag_field} {tag!r}. Registered: {sorted(registry)}")
    return registry[tag](**value)


# --------------------------------------------------------------------------- tensor arguments
class _BaseArgs(BaseModel):
    model_config = ConfigDict(arbitrary_types_allowed=True)

    dtype: QuantType | str = QuantType.QInt8
    symmetric: bool = False
    group_size: int | None = Field(default=None, description=">0: group quant, -1: channel quant, None: tensor quant")
    strategy: QuantizationStrategy | str | None = None
    scale_dtype: np.dtype = Field(default=np.dtype(np.float32))
    zp_dtype: np.dtype = Field(default=None, init=False)   # filled in by the model validator
    reduce_range: bool = False

    @field_validator("dtype", mode="before")
    def _coerce_dtype(cls, value):
        return QuantType.from_string(value) if isinstance(value, str) else value

    @field_validator("group_size", mode="before")
    def _check_group_size(cls, value):
        if value is not None and value < -1:
            raise ValueError(f"Invalid group size {value}. Use group_size > 0 for "
                             "strategy='group' and group_size = -1 for 'per_channel'")
        return value

    @field_validator("strategy", mode="before")
    def _coerce_strategy(cls, value):
        return QuantizationStrategy(value.lower()) if isinstance(value, str) else value

    @field_validator("scale_dtype", mode="before")
    def _coerce_scale_dtype(cls, value):
        return value if isinstance(value, np.dtype) else np.dtype(value)

    @field_validator("scale_dtype", mode="after")
    def _check_scale_dtype(cls, value):
        if value != np.float32:
            raise ValueError("Only float32 scale dtype is currently supported.")
        return value

    @model_validator(mode="after")
    def validate_model_after(self):
        strategy, g = self.strategy, self.group_size
        if strategy is None:   # infer from group_size: None -> tensor, -1 -> channel, > 0 -> group
            if g is None:
                strategy = QuantizationStrategy.TENSOR
            elif g == -1:
                strategy = QuantizationStrategy.CHANNEL
            elif g > 0:
                strategy = QuantizationStrategy.GROUP
            else:
                raise ValueError(f"Invalid group size {g}. Use group_size > 0 for "
                                 "strategy='group' and group_size = -1 for 'channel'")
        if strategy == QuantizationStrategy.GROUP and (g is None or g <= 0):
            raise ValueError(f"strategy {strategy} requires group_size to be set to a positive value.")
        if g is not None and g > 0 and strategy != QuantizationStrategy.GROUP:
            raise ValueError("group_size requires strategy to be set to 'group'.")
        if self.zp_dtype is None:
            self.zp_dtype = self.dtype.np_dtype
        self.strategy = strategy
        return self


class QWeightArgs(_BaseArgs):
    """Weight quantization parameters (reference: _qconfig.py:271-301).  Unknown keyword arguments are
    ignored, as in the reference (its rewrite rules rely on that)."""

    clip_ratio: float = 1.0
    mse: bool = False
    algorithm: SerializeAsAny[AlgorithmConfig] = Field(default_factory=_default_algorithm_config)

    @field_validator("algorithm", mode="before")
    def _coerce_algorithm(cls, value):
        if value is None:
            return _default_algorithm_config()
        return _from_registry(value, _ALGORITHM_REGISTRY, "algorithm_type", AlgorithmConfig)

    @field_validator("clip_ratio", mode="after")
    def _check_clip_ratio(cls, value):
        if not (0.0 < value <= 1.0):
            raise ValueError(f"clip_ratio must be in (0.0, 1.0], got {value}")
        return value

    @model_validator(mode="after")
    def validate_model_after(self):
        self.algorithm.validate_weight_args(self)
        return super().validate_model_after()


class QActivationArgs(_BaseArgs):
    """Activation quantization parameters (reference: _qconfig.py:304-335): per-tensor, 8-bit only."""

    is_static: bool = True

    @field_validator("strategy", mode="after")
    def _tensor_only(cls, value):
        if value is not None and value != QuantizationStrategy.TENSOR:
            raise NotImplementedError("Activation quantization only supports 'tensor' strategy.")
        return QuantizationStrategy.TENSOR

    @field_validator("dtype", mode="after")
    def _no_four_bit(cls, value):
        if value in _FOUR_BIT:
            raise NotImplementedError("4-bit quantization is not supported for activations.")
        return value

    @model_validator(mode="after")
    def validate_model_after(self):
        if not self.is_static and self.dtype != QuantType.QUInt8:
            raise NotImplementedError("Dynamic activation quantization only supports uint8 dtype.")
        return super().validate_model_after()


# --------------------------------------------------------------------------- top-level config
class QConfig(BaseModel):
    """Main configuration object (reference: _qconfig.py:338-502)."""

    model_config = ConfigDict(extra="forbid", arbitrary_types_allowed=True)

    target_op_types: Sequence[str] = Field(default_factory=lambda: _SUPPORTED_OP_TYPES)
    weights: QWeightArgs | None = None
    input_activations: QActivationArgs | None = None
    output_activations: QActivationArgs | None = None
    format: QFormat | str = QFormat.QDQ
    calibration_params: CalibrationParams | None = Field(default_factory=CalibrationParams)
    calibration_data: np.ndarray | dict[str, np.ndarray] | None = None
    preprocessors: Sequence[SerializeAsAny[PreProcessingConfig]] = Field(default_factory=tuple)
    ignore: Sequence[str] = Field(default_factory=tuple)

    @field_validator("target_op_types", mode="before")
    def _normalise_op_types(cls, value):
        return tuple(sorted(set(value)))

    @field_validator("ignore", mode="before")
    def _normalise_ignore(cls, value):
        if value is None:
            return ()
        return (value,) if isinstance(value, str) else tuple(value)

    @field_validator("preprocessors", mode="before")
    def _coerce_preprocessors(cls, value):
        if value is None:
            return ()
        return tuple(_from_registry(v, _PREPROCESSING_REGISTRY, "preprocessing_type", PreProcessingConfig)
                     for v in value)

    @field_validator("format", mode="before")
    def _coerce_format(cls, value):
        if isinstance(value, str):
            try:
                return QFormat(value.lower())
            except ValueError:
                raise ValueError(f"Invalid quantization format '{value}'. "  # noqa: B904
                                 f"Valid formats are: {[f.value for f in QFormat]}")
        return value

    @field_validator("calibration_params", mode="before")
    def _coerce_calibration_params(cls, value):
        return CalibrationParams(**value) if isinstance(value, dict) else value

    def _require_qlinear_compatible(self) -> None:
        acts = (("input", self.input_activations), ("output", self.output_activations))
        if any(a is None for _, a in acts):
            raise ValueError("QLinear format requires both input and output activation quantization.")
        if not all(a.is_static for _, a in acts):
            raise ValueError("QLinear format requires both input and output activations "
                             "quantization to be static.")
        if self.weights.strategy == QuantizationStrategy.GROUP:
            raise NotImplementedError("QLinear format does not support grouped weight quantization.")
        if self.weights.dtype not in _EIGHT_BIT:
            raise ValueError(f"QLinear format supports only int8/uint8 for weights, got {self.weights.dtype}.")
        for name, a in acts:
            if a.dtype not in _EIGHT_BIT:
                raise ValueError(f"QLinear format supports only int8/uint8 for {name} activations, got {a.dtype}.")

    @model_validator(mode="after")
    def validate_model_after(self):
        for op_type in self.target_op_types:
            if op_type not in _SUPPORTED_OP_TYPES:
                raise ValueError(f"Unsupported operator type '{op_type}' in target_op_types. "
                                 f"Supported operator types are: {_SUPPORTED_OP_TYPES}")
        acts = (self.input_activations, self.output_activations)
        if self.weights is None:
            if all(a is None for a in acts):
                return self                                   # nothing to quantize
            raise ValueError("Activation only quantization is not supported.")
        weights_only = all(a is None for a in acts)
        if not weights_only:
            if self.weights.dtype in _FOUR_BIT:
                raise NotImplementedError("4-bit quantization is only supported for weights_only quantization.")
            if self.weights.strategy == QuantizationStrategy.GROUP:
                raise NotImplementedError("Group quantization is only supported for weights_only quantization.")
        if all(a is not None for a in acts) and acts[0].is_static != acts[1].is_static:
            raise NotImplementedError("Both input and output activations must be either both static or dynamic.")
        if self.format == QFormat.QLINEAR:
            self._require_qlinear_compatible()
        return self
