"""Calibration parameters and the GPU min-max calibrator (reference: core/_calibration/{base,minmax,
factory}.py).  The calibrator keeps the running (min, max) of every tensor name in device memory and
updates it with one HBM-bound reduction per batch (oq_minmax_collect_f32/f64); nothing is copied back
until ``data[name]`` or ``compute_range`` is read.
"""
from __future__ import annotations

import abc
import enum
import logging
from typing import Any

import numpy as np
from pydantic import BaseModel, ConfigDict, Field, field_validator

__all__ = ["CalibrationMethod", "CalibrationParams"]

logger = logging.getLogger(__name__)


class ExecutionProvider(str, enum.Enum):
    """onnxruntime execution provider used for the calibration forward pass (base.py:12-32).  The two
    ROCm providers are additions for MI355X hosts; the reference's names and aliases are unchanged."""

    CPU = "CPUExecutionProvider"
    CUDA = "CUDAExecutionProvider"
    ROCM = "ROCMExecutionProvider"
    MIGRAPHX = "MIGraphXExecutionProvider"

    @classmethod
    def from_alias(cls, value: str) -> "ExecutionProvider":
        aliases = {"cpu": cls.CPU, "cuda": cls.CUDA, "gpu": cls.CUDA, "rocm": cls.ROCM, "migraphx": cls.MIGRAPHX}
        hit = aliases.get(value.lower())
        if hit is not None:
            return hit
        try:
            return cls(value)
        except ValueError:
            valid = sorted({*aliases, *(p.value for p in cls)})
            raise ValueError(f"Invalid execution provider '{value}'. Valid values are: {valid}")  # noqa: B904


class CalibrationMethod(enum.Enum):
    MINMAX = "minmax"


class CalibrationParams(BaseModel):
    """base.py:41-97: method, num_samples=100, batch_size=10, momentum=0.0, provider=CPU; extra keys forbidden."""

    model_config = ConfigDict(extra="forbid")

    method: CalibrationMethod | str = CalibrationMethod.MINMAX
    num_samples: int = 100
    batch_size: int = 10
    momentum: float = 0.0
    provider: ExecutionProvider | str = Field(default=ExecutionProvider.CPU)

    @field_validator("method", mode="before")
    def _coerce_method(cls, value):
        if isinstance(value, str):
            try:
                return CalibrationMethod(value)
            except ValueError:
                raise ValueError(f"Invalid calibration method '{value}'. "  # noqa: B904
                                 f"Valid methods are: {[m.value for m in CalibrationMethod]}")
        return value

    @field_validator("provider", mode="before")
    def _coerce_provider(cls, value):
        return ExecutionProvider.from_alias(value) if isinstance(value, str) else value

    @field_validator("momentum", mode="after")
    def _check_momentum(cls, value):
        if not 0 <= value < 1:
            raise ValueError(f"Momentum must be in [0, 1), got {value}")
        return value

    @field_validator("num_samples", "batch_size", mode="after")
    def _check_positive(cls, value, info):
        if value <= 0:
            raise ValueError(f"{info.field_name} must be positive, got {value}")
        return value


class CalibrationData:
    """Statistics of one tensor name (base.py:100-110).  ``min_val`` / ``max_val`` are read from the
    device-resident state on access (this is the only host synchronisation of the calibrator)."""

    __slots__ = ("_state", "_np_dtype")

    def __init__(self, state, np_dtype):
        self._state = state            # torch tensor [4] on the GPU: {min, max, seen, -}
        self._np_dtype = np_dtype

    def _read(self, i: int):
        return self._np_dtype.type(self._state[i].item())

    @property
    def min_val(self):
        return self._read(0)

    @property
    def max_val(self):
        return self._read(1)

    def __repr__(self) -> str:  # pragma: no cover
        return f"CalibrationData(min_val={self.min_val}, max_val={self.max_val})"


class Calibrator(abc.ABC):
    """base.py:113-144."""

    def __init__(self):
        self.data: dict[str, CalibrationData] = {}

    @abc.abstractmethod
    def collect(self, name: str, array) -> None: ...

    @abc.abstractmethod
    def compute_range(self, name: str) -> tuple[np.ndarray, np.ndarray]: ...


class MinMaxCalibrator(Calibrator):
    """minmax.py:11-87 on the GPU.  ``collect`` accepts a NumPy array (copied to HBM) or a torch tensor
    already in HBM (no copy: the on-device calibration driver of SURVEY.md 8f-N1 feeds those)."""

    def __init__(self, momentum: float = 0.0):
        super().__init__()
        assert 0 <= momentum < 1, "Momentum must be in the range [0, 1)."
        self.momentum = momentum
        logger.debug(f"Initialized MinMaxCalibrator with momentum={momentum}")

    def collect(self, name: str, array) -> None:
        import torch

        from .hip import ops

        if isinstance(array, np.ndarray):
            if array.dtype not in (np.float32, np.float64):
                if array.dtype == np.float16:
                    array = array.astype(np.float32)
                else:
                    raise TypeError(f"MinMaxCalibrator (HIP) supports float32/float64 activations, got {array.dtype}")
            x = torch.from_numpy(np.ascontiguousarray(array)).cuda()
        elif isinstance(array, torch.Tensor):
            x = array if array.dtype in (torch.float32, torch.float64) else array.to(torch.float32)
            if not x.is_cuda:
                x = x.cuda()
        else:
            raise TypeError("collect() expects a numpy array or a torch tensor")
        if name not in self.data:
            np_dtype = np.dtype(np.float64 if x.dtype == torch.float64 else np.float32)
            self.data[name] = CalibrationData(ops.minmax_state(x.device, x.dtype), np_dtype)
        entry = self.data[name]
        if entry._state.dtype != x.dtype:
            x = x.to(entry._state.dtype)
        ops.minmax_collect(x, entry._state, self.momentum)

    def collect_many(self, arrays) -> None:
        """``collect`` for a whole calibration batch ({name: fp32 torch tensor in HBM}) in one launch pair -- what an
        on-device calibration driver calls once per batch instead of calibrate.py:264-266's per-tensor loop.  Same
        statistics as calling ``collect`` for every item (first sight / EMA / running min-max per name)."""
        import torch

        from .hip import ops

        names, xs = [], []
        for name, t in arrays.items():
            if not (isinstance(t, torch.Tensor) and t.is_cuda and t.dtype == torch.float32) or (
                    name in self.data and self.data[name]._state.dtype != torch.float32):
                self.collect(name, t)            # anything else takes the per-tensor path
                continue
            if name not in self.data:
                self.data[name] = CalibrationData(ops.minmax_state(t.device, torch.float32), np.dtype(np.float32))
            names.append(name)
            xs.append(t)
        if xs:
            ops.minmax_collect_many(xs, [self.data[n]._state for n in names], self.momentum)

    def compute_range(self, name: str) -> tuple[np.ndarray, np.ndarray]:
        if name not in self.data:
            raise KeyError(f"No calibration data collected for '{name}'")
        entry = self.data[name]
        lo, hi = entry._state[:2].tolist()                      # one D2H copy for both values
        lo, hi = entry._np_dtype.type(lo), entry._np_dtype.type(hi)
        # minmax.py:83-87 zero is always part of the range
        return np.array(np.minimum(lo, 0), dtype=np.float32), np.array(np.maximum(hi, 0), dtype=np.float32)


    def compute_qparams_many(self, names, quant_type, is_symmetric: bool = False, reduce_range: bool = False):
        """``compute_range`` + ``_compute_qparams`` (calibrate.py:268-285) for a list of fp32 names with ONE kernel and
        ONE device-to-host copy instead of a round trip per name.  Returns {name: (scale fp32 0-d, zero_point 0-d in the
        activation dtype)}; unseen names raise KeyError like ``compute_range``."""
        import torch

        from .hip import ops

        names = list(names)
        for n in names:
            if n not in self.data:
                raise KeyError(f"No calibration data collected for '{n}'")
        if not names:
            return {}
        if any(self.data[n]._state.dtype != torch.float32 for n in names):     # float64 names: the per-name path
            from .algorithms.functional import _compute_qparams
            return {n: _compute_qparams(*self.compute_range(n), quant_type, is_symmetric, reduce_range, np.float32, quant_type.np_dtype)
                    for n in names}
        st = torch.stack([self.data[n]._state for n in names])               # [n, 4] on the device
        zero = torch.zeros((), dtype=torch.float32, device=st.device)
        lo, hi = torch.minimum(st[:, 0], zero), torch.maximum(st[:, 1], zero)  # minmax.py:83-87 zero is part of the range
        scale, zp = ops.qparams(lo.contiguous(), hi.contiguous(), quant_type.key, bool(is_symmetric), bool(reduce_range))
        both = torch.stack([scale, zp.to(torch.float32)]).cpu().numpy()       # zero points of 8-bit grids are exact in fp32
        return {n: (np.array(both[0, i], dtype=np.float32), np.array(both[1, i]).astype(quant_type.np_dtype))
                for i, n in enumerate(names)}


_CALIBRATORS: dict[CalibrationMethod, type[Calibrator]] = {CalibrationMethod.MINMAX: MinMaxCalibrator}


def get_calibrator(method: CalibrationMethod = CalibrationMethod.MINMAX, **kwargs: Any) -> Calibrator:
    """factory.py:15-32."""
    calibrator_class = _CALIBRATORS[method]
    try:
        return calibrator_class(**kwargs)
    except TypeError as e:
        raise TypeError(f"Invalid arguments for {calibrator_class.__name__}: {e}") from e
