"""Quantization value types (reference: core/_dtypes.py).

``QuantType`` members carry an ONNX element type.  When ``onnx_ir`` is installed its
``DataType`` is used, exactly like the reference; otherwise a small stand-in with the same
ONNX codes is used so that the numeric API works without the ONNX stack.  4-bit values are
held one per byte (``ml_dtypes.int4/uint4`` when that package is present, else int8/uint8
containers with identical values).
"""
from __future__ import annotations

import enum

import numpy as np

__all__ = ["QuantType"]

try:  # pragma: no cover - depends on the environment
    import onnx_ir as _ir

    _DataType = _ir.DataType
except Exception:  # onnx_ir absent: same names, same ONNX TensorProto codes
    class _DataType(enum.IntEnum):
        UINT8 = 2
        INT8 = 3
        INT32 = 6
        UINT32 = 12
        UINT4 = 21
        INT4 = 22

        def numpy(self) -> np.dtype:
            return _NP_FALLBACK[self.name]

        @property
        def bitwidth(self) -> int:
            return 4 if self.name.endswith("4") else (8 if self.name.endswith("8") else 32)

_NP_FALLBACK = {"UINT8": np.dtype(np.uint8), "INT8": np.dtype(np.int8), "INT32": np.dtype(np.int32),
                "UINT32": np.dtype(np.uint32), "UINT4": np.dtype(np.uint8), "INT4": np.dtype(np.int8)}

# name, bits, signed -> everything else is derived.  core/_dtypes.py:8-30 as arithmetic instead of tables:
#   full range      signed [-2^(b-1), 2^(b-1)-1]   unsigned [0, 2^b - 1]
#   symmetric       signed only: [-(2^(b-1)-1), 2^(b-1)-1]
#   reduced range   signed [-2^(b-2), 2^(b-2)] for 8/32 bits but [-4, 3] for int4;  unsigned [0, 2^(b-1)-1]
_KINDS = {"int4": (4, True), "uint4": (4, False), "int8": (8, True), "uint8": (8, False),
          "int32": (32, True), "uint32": (32, False)}


def _qrange(bits: int, signed: bool, symmetric: bool, reduce_range: bool) -> tuple[int, int]:
    if reduce_range:                                      # wins over symmetric (core/_dtypes.py:63-64)
        if not signed:
            return 0, 2 ** (bits - 1) - 1
        if bits == 4:
            return -4, 3
        return -(2 ** (bits - 2)), 2 ** (bits - 2)
    if signed:
        hi = 2 ** (bits - 1) - 1
        return (-hi, hi) if symmetric else (-hi - 1, hi)
    return 0, 2 ** bits - 1                               # unsigned types have no symmetric entry


class QuantType(enum.Enum):
    """Enumeration of quantization types (same member names / values as the reference)."""

    QInt4 = _DataType.INT4
    QUInt4 = _DataType.UINT4
    QInt8 = _DataType.INT8
    QUInt8 = _DataType.UINT8
    QInt32 = _DataType.INT32
    QUInt32 = _DataType.UINT32

    @classmethod
    def from_string(cls, value: str) -> "QuantType":
        key = value.lower().strip()
        if key not in _KINDS:
            raise ValueError(f"Invalid quantization type '{value}'. Expected one of: {', '.join(_KINDS)}")
        return cls[("QUInt" if key.startswith("u") else "QInt") + key.lstrip("uint")]

    @property
    def key(self) -> str:
        """Lower-case tag used by the HIP layer ("int4", "uint8", ...)."""
        return self.name[1:].lower()

    @property
    def np_dtype(self) -> np.dtype:
        return self.value.numpy()

    @property
    def bitwidth(self) -> int:
        return self.value.bitwidth

    @property
    def signed(self) -> bool:
        return _KINDS[self.key][1]

    def qrange(self, is_symmetric: bool, reduce_range: bool = False) -> tuple[int, int]:
        bits, signed = _KINDS[self.key]
        return _qrange(bits, signed, bool(is_symmetric), bool(reduce_range))
