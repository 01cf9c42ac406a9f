"""ctypes binding of liboq_hip.so -- the C-ABI boundary declared in include/oq_hip.h.

This is the stub a maintainer of the reference would add to call the HIP path from
NumPy-land (INTEGRATION.md shows it next to the reference call sites it replaces).
There is NO CPU fallback: if the shared library is missing or a call fails, an
exception is raised.
"""
from __future__ import annotations

import ctypes as C
import os
import threading

_PKG = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB_PATH = os.path.join(_PKG, "lib", "liboq_hip.so")

# enums of include/oq_hip.h
OQ_INT4, OQ_UINT4, OQ_INT8, OQ_UINT8, OQ_INT32, OQ_UINT32 = range(6)
OQ_TENSOR, OQ_CHANNEL, OQ_GROUP = range(3)
OQ_LAYOUT_KN, OQ_LAYOUT_NBITS, OQ_LAYOUT_KN_PACKED4 = range(3)
OQ_GPTQ_PARITY, OQ_GPTQ_CORRECTED, OQ_GPTQ_CORRECTED_COLUMNS = range(3)
OQ_ABI_VERSION = 2
OQ_ERR_INVALID_ARGUMENT, OQ_ERR_UNSUPPORTED, OQ_ERR_WORKSPACE, OQ_ERR_LAUNCH, OQ_ERR_NOT_SPD = -1, -2, -3, -4, -5

QTYPE_CODE = {"int4": OQ_INT4, "uint4": OQ_UINT4, "int8": OQ_INT8, "uint8": OQ_UINT8,
              "int32": OQ_INT32, "uint32": OQ_UINT32}
STRATEGY_CODE = {"tensor": OQ_TENSOR, "channel": OQ_CHANNEL, "group": OQ_GROUP}


class OqHipError(RuntimeError):
    """A liboq_hip call returned a negative status."""

    def __init__(self, status: int, message: str):
        super().__init__(f"liboq_hip status {status}: {message}")
        self.status = status
        self.message = message


class OqHipMissing(ImportError):
    """The HIP extension is not built / not loadable.  Never silently replaced by a CPU path."""


_i32, _i64, _f32, _f64 = C.c_int32, C.c_int64, C.c_float, C.c_double
_p, _sz = C.c_void_p, C.c_size_t

# name -> (restype, argtypes); mirrors include/oq_hip.h one to one (tests check the set).
PROTOTYPES = {
    "oq_abi_version": (_i32, []),
    "oq_last_error": (C.c_char_p, []),
    "oq_status_string": (C.c_char_p, [_i32]),
    "oq_target_arch": (C.c_char_p, []),
    "oq_qrange": (_i32, [_i32, _i32, _i32, C.POINTER(_i64), C.POINTER(_i64)]),
    "oq_rtn_workspace_bytes": (_sz, [_i64, _i64, _i32, _i64, _i32]),
    "oq_rtn_quantize_f32": (_i32, [_p, _i64, _i64, _i64, _i32, _i32, _i64, _i32, _i32, _f32, _i32,
                                   _p, _p, _p, _i32, _p, _sz, _p]),
    "oq_rtn_batched_workspace_bytes": (_sz, [_i64, _i64, _i64, _i64]),
    "oq_rtn_quantize_batched_f32": (_i32, [_p, _i64, _i64, _i64, _i64, _i64, _i32, _i64, _i32, _i32, _f32, _p, _p, _p,
                                           _i32, _p, _sz, _p]),
    "oq_hessian_many_workspace_bytes": (_sz, [_p, _i64]),
    "oq_hessian_accumulate_many_f32": (_i32, [_p, _p, _i64, _p, _sz, _p]),
    "oq_hessian_pieces_bytes": (_sz, [_i64, _i64]),
    "oq_hessian_slab_bytes": (_sz, [_i64]),
    "oq_hessian_prepare_f32": (_i32, [_p, _i64, _i64, _i64, _i64, _p, _sz, _p]),
    "oq_hessian_accumulate_prepared_f32": (_i32, [_p, _i64, _i64, _i64, _i64, _p, _p, _sz, _p]),
    "oq_matmul_pieces_bytes": (_sz, [_i64, _i64]),
    "oq_matmul_prepare_f32": (_i32, [_p, _i64, _i64, _i64, _i32, _i32, _p, _sz, _p]),
    "oq_matmul_pieces_f32": (_i32, [_p, _p, _i64, _i64, _i64, _f32, _f32, _p, _i64, _i32, _p]),
    "oq_awq_workspace_bytes": (_sz, [_i64, _i64, _i64]),
    "oq_awq_scale_search_f32": (_i32, [_p, _i64, _i64, _i64, _p, _i64, _i64, _i32, _i32, _i64, _i32, _i32, _i32, _p, _p, _p, _p, _sz, _p]),
    "oq_awq_clip_search_f32": (_i32, [_p, _i64, _i64, _i64, _p, _i64, _i64, _i32, _i32, _i64, _i32, _i32, _p, _p, _p, _sz, _p]),
    "oq_awq_stats_workspace_bytes": (_sz, [_i64, _i64]),
    "oq_awq_scale_search_stats_f32": (_i32, [_p, _p, _i64, _i64, _p, _i64, _i64, _i32, _i32, _i64, _i32, _i32, _i32, _p, _p, _p, _p, _sz, _p]),
    "oq_awq_clip_search_stats_f32": (_i32, [_p, _i64, _i64, _p, _i64, _i64, _i32, _i32, _i64, _i32, _i32, _p, _p, _p, _sz, _p]),
    "oq_abs_sum_cols_workspace_bytes": (_sz, [_i64]),
    "oq_abs_sum_cols_f32": (_i32, [_p, _i64, _i64, _i64, _p, _i32, _p, _sz, _p]),
    "oq_fingerprint64": (_i32, [_p, _i64, _p, _p]),
    "oq_smooth_quant_workspace_bytes": (_sz, [_i64]),
    "oq_smooth_quant_scale_f32": (_i32, [_p, _i64, _i64, _i64, _p, _i64, _i64, _f32, _p, _p, _sz, _p]),
    "oq_rtn_state_bytes": (_sz, [_i64, _i64, _i32, _i64]),
    "oq_rtn_quantize_stateful_f32": (_i32, [_p, _i64, _i64, _i64, _i32, _i32, _i64, _i32, _i32, _f32, _i32, _p, _p, _p, _i32, _p, _sz, _p, _sz, _p]),
    "oq_rtn_quantize_ptrs_f32": (_i32, [_p, _p, _i64, _i64, _i64, _i64, _i32, _i64, _i32, _i32, _f32, _i32, _p, _sz, _p]),
    "oq_rtn_qparams_f32": (_i32, [_p, _i64, _i64, _i64, _i32, _i32, _i64, _i32, _i32, _f32, _i32,
                                  _p, _p, _p, _sz, _p]),
    "oq_qparams_f32": (_i32, [_p, _p, _i64, _i32, _i32, _i32, _p, _p, _p]),
    "oq_qparams_f64": (_i32, [_p, _p, _i64, _i32, _i32, _i32, _p, _p, _p]),
    "oq_minmax_rows_f32": (_i32, [_p, _i64, _i64, _i64, _p, _p, _p]),
    "oq_quantize_f32": (_i32, [_p, _i64, _i64, _i64, _p, _p, _i64, _i64, _i64, _i32, _i32, _i32, _p, _p]),
    "oq_dequantize_f32": (_i32, [_p, _i64, _i64, _i32, _p, _p, _i64, _i64, _i64, _p, _i64, _p]),
    "oq_dequantize_fzp_f32": (_i32, [_p, _i64, _i64, _i32, _p, _p, _i64, _i64, _i64, _p, _i64, _p]),
    "oq_quantize_bias_f32": (_i32, [_p, _i64, _p, _i64, _f32, _p, _p, _p]),
    "oq_minmax_workspace_bytes": (_sz, [_i64]),
    "oq_minmax_collect_f32": (_i32, [_p, _i64, _p, _f64, _p, _sz, _p]),
    "oq_minmax_collect_f64": (_i32, [_p, _i64, _p, _f64, _p, _sz, _p]),
    "oq_minmax_many_workspace_bytes": (_sz, [_i64]),
    "oq_minmax_collect_many_f32": (_i32, [_p, _i64, _f64, _p, _sz, _p]),
    "oq_absmax_workspace_bytes": (_sz, [_i64, _i64, _i32]),
    "oq_absmax_f32": (_i32, [_p, _i64, _i64, _i64, _i32, _p, _p, _sz, _p]),
    "oq_rtn_tensor_many_workspace_bytes": (_sz, [_i64]),
    "oq_rtn_tensor_many_f32": (_i32, [_p, _i64, _i32, _i32, _i32, _f32, _p, _sz, _p]),
    "oq_pack_matmul_nbits": (_i32, [_p, _i64, _i64, _i64, _i32, _p, _p]),
    "oq_hessian_workspace_bytes": (_sz, [_i64, _i64]),
    "oq_hessian_accumulate_f32": (_i32, [_p, _i64, _i64, _i64, _i64, _i64, _p, _i32, _p, _sz, _p]),
    "oq_gptq_prepare_workspace_bytes": (_sz, [_i64, _i64, _i32]),
    "oq_gptq_prepare_f32": (_i32, [_p, _i64, _i64, _p, _i32, _p, _p, _sz, _p]),
    "oq_gptq_factor_workspace_bytes": (_sz, [_i64]),
    "oq_gptq_factor_f32": (_i32, [_p, _i64, _f32, _p, _p, _i32, _p, _sz, _p]),
    "oq_gptq_factor_batched_workspace_bytes": (_sz, [_i64, _i64]),
    "oq_gptq_factor_batched_f32": (_i32, [_p, _i64, _i64, _i64, _f32, _i32, _p, _i64, _p, _i32, _p, _sz, _p]),
    "oq_gptq_loop_workspace_bytes": (_sz, [_i64, _i64, _i64]),
    "oq_gptq_loop_f32": (_i32, [_p, _i64, _i64, _p, _i32, _i64, _i32, _i32, _f32, _i32, _i64, _i32, _i32,
                                _p, _p, _i64, _p, _i32, _p, _p, _p, _p, _sz, _p]),
    "oq_hqq_workspace_bytes": (_sz, [_i64, _i64, _i64]),
    "oq_hqq_optimize_f32": (_i32, [_p, _i64, _i64, _i64, _i64, _i32, _p, _p, _f64, _f64, _f64, _i32, _i32, _i32, _p, _i32, _p, _p,
                                   _p, _sz, _p]),
    "oq_pack_zero_points_u4": (_i32, [_p, _i64, _i64, _p, _p]),
    "oq_pack_nibbles": (_i32, [_p, _i64, _p, _p]),
}

_lock = threading.Lock()
_lib = None


def load() -> C.CDLL:
    """Load liboq_hip.so once; raise OqHipMissing (loudly) when it is not there."""
    global _lib
    if _lib is not None:
        return _lib
    with _lock:
        if _lib is not None:
            return _lib
        if not os.path.exists(LIB_PATH):
            raise OqHipMissing(
                f"{LIB_PATH} not found: build it with `python -m onnx_quantize_amd._build` "
                "(hipcc, gfx950).  onnx_quantize_amd has no CPU fallback.")
        try:
            # torch ships its own libamdhip64; it has to be the HIP runtime of the process, otherwise the
            # library's launches go to a second runtime instance that never saw torch's device context
            # ("no ROCm-capable device is detected").  Importing torch first makes the dynamic linker
            # resolve liboq_hip's libamdhip64 dependency to the copy torch already loaded.
            import torch  # noqa: F401

            lib = C.CDLL(LIB_PATH)
        except OSError as e:  # missing ROCm runtime etc.
            raise OqHipMissing(f"cannot load {LIB_PATH}: {e}") from e
        for name, (res, args) in PROTOTYPES.items():
            try:
                fn = getattr(lib, name)
            except AttributeError as e:
                raise OqHipMissing(f"{LIB_PATH} does not export {name}; rebuild it") from e
            fn.restype = res
            fn.argtypes = args
        if lib.oq_abi_version() != OQ_ABI_VERSION:
            raise OqHipMissing(f"{LIB_PATH} has ABI {lib.oq_abi_version()}, expected {OQ_ABI_VERSION}")
        _lib = lib
    return _lib


def check(status: int) -> None:
    if status != 0:
        msg = load().oq_last_error().decode("utf-8", "replace")
        raise OqHipError(status, msg)


def qrange(qtype: int, symmetric: bool, reduce_range: bool) -> tuple[int, int]:
    lo, hi = _i64(), _i64()
    check(load().oq_qrange(qtype, int(symmetric), int(reduce_range), C.byref(lo), C.byref(hi)))
    return lo.value, hi.value
