#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ FROM THE REFERENCE ITSELF.

Runs only in the build container (needs /root/reference).  It imports the reference's
numeric modules -- and nothing else of the reference -- through the two stand-ins
described in SURVEY.md section 8c:

  * a module ``onnx_ir`` that only offers ``DataType`` (an IntEnum with ``numpy()`` and
    ``bitwidth``), because ``onnx_ir`` is not installed here;
  * an empty package object ``onnx_quantize`` whose ``__path__`` points at the reference
    sources, so ``onnx_quantize/__init__.py`` (which pulls in onnx / onnxscript) is
    bypassed.

Only DATA leaves this script: inputs, parameters and the arrays the reference returned,
as ``.npz`` / ``.json``.  No reference source, bytecode or pickled object is written.

    python tests/golden/make_golden.py            # rewrites tests/golden/*.npz|json
"""

from __future__ import annotations

import enum
import hashlib
import importlib
import itertools
import json
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference/src/onnx_quantize"


def _load_reference():
    sys.dont_write_bytecode = True

    class DataType(enum.IntEnum):
        UINT8 = 2
        INT8 = 3
        INT32 = 6
        INT64 = 7
        UINT32 = 12
        UINT4 = 21
        INT4 = 22

        def __str__(self):                      # onnx_ir.DataType prints its name; the config serialiser relies on it
            return self.name

        def numpy(self):
            return {
                2: np.dtype(np.uint8), 3: np.dtype(np.int8), 6: np.dtype(np.int32), 7: np.dtype(np.int64),
                12: np.dtype(np.uint32), 21: np.dtype(np.uint8), 22: np.dtype(np.int8),
            }[int(self)]

        @property
        def bitwidth(self):
            return {2: 8, 3: 8, 6: 32, 12: 32, 21: 4, 22: 4}[int(self)]

    ir = types.ModuleType("onnx_ir")
    ir.DataType = DataType
    sys.modules["onnx_ir"] = ir
    pkg = types.ModuleType("onnx_quantize")
    pkg.__path__ = [REF]
    sys.modules["onnx_quantize"] = pkg

    mods = types.SimpleNamespace()
    mods.utils = importlib.import_module("onnx_quantize.core._algorithms.utils")
    mods.rtn = importlib.import_module("onnx_quantize.core._algorithms.rtn")
    mods.gptq = importlib.import_module("onnx_quantize.core._algorithms.gptq")
    mods.hqq = importlib.import_module("onnx_quantize.core._algorithms.hqq")
    mods.minmax = importlib.import_module("onnx_quantize.core._calibration.minmax")
    mods.pack = importlib.import_module("onnx_quantize.core._pack")
    mods.dtypes = importlib.import_module("onnx_quantize.core._dtypes")
    mods.qconfig = importlib.import_module("onnx_quantize.core._qconfig")
    return mods


R = _load_reference()
QT = {
    "int4": R.dtypes.QuantType.QInt4, "uint4": R.dtypes.QuantType.QUInt4,
    "int8": R.dtypes.QuantType.QInt8, "uint8": R.dtypes.QuantType.QUInt8,
    "int32": R.dtypes.QuantType.QInt32, "uint32": R.dtypes.QuantType.QUInt32,
}
ST = {
    "tensor": R.qconfig.QuantizationStrategy.TENSOR,
    "channel": R.qconfig.QuantizationStrategy.CHANNEL,
    "group": R.qconfig.QuantizationStrategy.GROUP,
}


def sha16(a) -> str:
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()[:16]


def weight(kind: str, seed: int, k: int, n: int) -> np.ndarray:
    """Synthetic fp32 weights; the same recipe is re-run by the tests for the big cases."""
    rng = np.random.default_rng(seed)
    if kind == "normal":
        return rng.standard_normal((k, n), dtype=np.float32)
    if kind == "heavy":
        return rng.standard_t(3, size=(k, n)).astype(np.float32)
    if kind == "positive":
        return rng.random((k, n), dtype=np.float32) + np.float32(0.25)
    if kind == "negative":
        return -(rng.random((k, n), dtype=np.float32) + np.float32(0.25))
    if kind == "zero_groups":
        w = rng.standard_normal((k, n), dtype=np.float32)
        w[: k // 2, ::3] = 0.0            # whole groups of zeros -> tiny-scale guard
        w[:, 1] = 0.0                     # a whole dead output channel
        return w
    if kind == "tiny":
        return (rng.standard_normal((k, n), dtype=np.float32) * np.float32(1e-39)).astype(np.float32)
    if kind == "halves":
        # values that land exactly on .5 after division -> exercises ties-to-even
        return (rng.integers(-31, 32, size=(k, n)).astype(np.float32)) * np.float32(0.5)
    raise ValueError(kind)


def rtn_call(w, qtype, strategy, g, sym, red, clip, mse):
    q, s, z = R.rtn._rtn_quantize(
        w, QT[qtype], ST[strategy], g, sym, red, clip, mse,
        np.dtype(np.float32), QT[qtype].np_dtype)
    return np.ascontiguousarray(q), np.asarray(s), np.asarray(z)


def gen_rtn_small(out):
    """Every dtype x strategy x symmetric x reduce_range on <=64x48 matrices."""
    cases = []
    arrays = {}
    shapes = [(64, 48), (32, 20), (48, 7)]
    kinds = ["normal", "heavy", "zero_groups", "positive", "negative", "halves", "tiny"]
    idx = 0
    for qtype in ("int4", "uint4", "int8", "uint8"):
        for strategy, g in (("tensor", -1), ("channel", -1), ("group", 16), ("group", 8),
                            ("group", 32), ("group", 128), ("group", -1)):
            for sym in (False, True):
                for red in (False, True):
                    k, n = shapes[idx % len(shapes)]
                    if strategy == "group" and g > 0 and k % min(g, k):
                        k = 64
                    kind = kinds[idx % len(kinds)]
                    clip = (1.0, 0.9, 0.75)[idx % 3]
                    seed = 100 + idx
                    w = weight(kind, seed, k, n)
                    q, s, z = rtn_call(w, qtype, strategy, g, sym, red, clip, False)
                    cid = f"c{idx:03d}"
                    arrays[f"{cid}_w"] = w
                    arrays[f"{cid}_q"] = q
                    arrays[f"{cid}_s"] = s
                    arrays[f"{cid}_z"] = z
                    cases.append(dict(id=cid, qtype=qtype, strategy=strategy, group_size=g,
                                      symmetric=sym, reduce_range=red, clip_ratio=clip,
                                      mse=False, kind=kind, seed=seed, k=k, n=n))
                    idx += 1
    # ragged / odd shapes: group that straddles columns (N*K % g == 0 but K % g != 0),
    # g = 1, single row, single column
    extra = [
        ("uint4", "group", 8, False, False, 12, 6, "normal"),
        ("int8", "group", 1, False, False, 8, 5, "normal"),
        ("int4", "group", 4, True, False, 4, 9, "heavy"),
        ("uint8", "channel", -1, False, False, 1, 33, "normal"),
        ("int8", "tensor", -1, True, False, 33, 1, "normal"),
        ("uint4", "group", 16, False, False, 256, 3, "normal"),
        ("int4", "group", 64, False, False, 128, 130, "heavy"),
        ("uint8", "group", 256, True, False, 512, 66, "normal"),
    ]
    for qtype, strategy, g, sym, red, k, n, kind in extra:
        seed = 100 + idx
        w = weight(kind, seed, k, n)
        q, s, z = rtn_call(w, qtype, strategy, g, sym, red, 1.0, False)
        cid = f"c{idx:03d}"
        arrays[f"{cid}_w"] = w
        arrays[f"{cid}_q"] = q
        arrays[f"{cid}_s"] = s
        arrays[f"{cid}_z"] = z
        cases.append(dict(id=cid, qtype=qtype, strategy=strategy, group_size=g, symmetric=sym,
                          reduce_range=red, clip_ratio=1.0, mse=False, kind=kind, seed=seed,
                          k=k, n=n))
        idx += 1
    np.savez_compressed(os.path.join(out, "rtn_small.npz"), **arrays)
    with open(os.path.join(out, "rtn_small.json"), "w") as f:
        json.dump(cases, f, indent=0)
    print("rtn_small:", len(cases), "cases")


def gen_rtn_mse(out):
    cases, arrays = [], {}
    grid = [
        ("uint4", "group", 32, False, "normal", 64, 24),
        ("int4", "group", 16, True, "heavy", 64, 24),
        ("int8", "channel", -1, False, "heavy", 48, 20),
        ("uint8", "tensor", -1, False, "normal", 32, 16),
        ("int8", "tensor", -1, True, "heavy", 32, 16),
        ("uint4", "group", 128, False, "heavy", 256, 16),
    ]
    for idx, (qtype, strategy, g, sym, kind, k, n) in enumerate(grid):
        seed = 700 + idx
        w = weight(kind, seed, k, n)
        q, s, z = rtn_call(w, qtype, strategy, g, sym, False, 1.0, True)
        rows = R.utils._preprocess_array(w, ST[strategy], g)
        lo, hi = R.utils._compute_min_max_mse(
            rows, QT[qtype], ST[strategy], g, sym, False, np.dtype(np.float32),
            QT[qtype].np_dtype)
        cid = f"m{idx:02d}"
        arrays.update({f"{cid}_w": w, f"{cid}_q": q, f"{cid}_s": s, f"{cid}_z": z,
                       f"{cid}_lo": np.asarray(lo), f"{cid}_hi": np.asarray(hi)})
        cases.append(dict(id=cid, qtype=qtype, strategy=strategy, group_size=g, symmetric=sym,
                          reduce_range=False, clip_ratio=1.0, mse=True, kind=kind, seed=seed,
                          k=k, n=n))
    np.savez_compressed(os.path.join(out, "rtn_mse.npz"), **arrays)
    with open(os.path.join(out, "rtn_mse.json"), "w") as f:
        json.dump(cases, f, indent=0)
    print("rtn_mse:", len(cases), "cases")


def gen_kernels(out):
    """quantize / dequantize / fake-quantize / bias / qparams on explicit inputs."""
    arrays = {}
    rng = np.random.default_rng(11)
    x = rng.standard_normal((40, 24), dtype=np.float32) * np.float32(3.0)
    for qtype in ("int4", "uint4", "int8", "uint8"):
        for sym in (False, True):
            for red in (False, True):
                tag = f"{qtype}_{int(sym)}{int(red)}"
                lo = np.minimum(x.min(axis=1, keepdims=True), 0)
                hi = np.maximum(x.max(axis=1, keepdims=True), 0)
                s, z = R.utils._compute_qparams(lo, hi, QT[qtype], sym, red,
                                                np.dtype(np.float32), QT[qtype].np_dtype)
                q = R.utils._quantize_array_from_qparams(x, s, z, QT[qtype], sym, red)
                dq = R.utils._dequantize_array(q, s, z)
                arrays[f"qp_{tag}_s"] = s
                arrays[f"qp_{tag}_z"] = z
                arrays[f"qp_{tag}_q"] = q
                arrays[f"qp_{tag}_dq"] = dq
    arrays["x"] = x
    # dequantize with preprocess=True in the three layouts
    w = weight("normal", 12, 64, 24)
    for strategy, g in (("tensor", -1), ("channel", -1), ("group", 16)):
        q, s, z = rtn_call(w, "uint8", strategy, g, False, False, 1.0, False)
        dq = R.utils._dequantize_array(q, s, z, preprocess=True, strategy=ST[strategy],
                                       group_size=g)
        arrays[f"dq_{strategy}_q"] = q
        arrays[f"dq_{strategy}_s"] = s
        arrays[f"dq_{strategy}_z"] = z
        arrays[f"dq_{strategy}_out"] = np.ascontiguousarray(dq)
    arrays["dq_w"] = w
    # bias
    bias = rng.standard_normal(48, dtype=np.float32)
    wscale = (rng.random(48, dtype=np.float32) + np.float32(0.01)) * np.float32(0.02)
    qb, bs, _ = R.rtn._quantize_bias(bias, np.float32(0.0371), wscale)
    arrays.update(bias=bias, bias_wscale=wscale, bias_xscale=np.float32(0.0371), bias_q=qb,
                  bias_scale=bs)
    np.savez_compressed(os.path.join(out, "kernels.npz"), **arrays)
    print("kernels: ok")


def gen_scalar_kats(out):
    """Known answers transcribed from the reference's own tests (data only) and checked
    here against the reference implementation before being written."""
    qparam_kats = [
        # (values, qtype, symmetric, expected_scale, expected_zp)  test_rtn.py:21-40
        ([0.0, 0.0, 0.0], "int8", False, 1.0, -128),
        ([0.0, 0.0, 0.0], "int8", True, 1.0, 0),
        ([0.0, 0.0, 0.0], "uint8", False, 1.0, 0),
        ([0.0, 0.0, 5.0], "int8", False, 5.0 / 255, -128),
        ([0.0, 0.0, 5.0], "int8", True, 10.0 / 254, 0),
        ([-5.0, -2.0, 0.0], "int8", False, 5.0 / 255, 127),
        ([-5.0, -2.0, 0.0], "int8", True, 5.0 / 127, 0),
        ([-5.0, 0.0, 5.0], "int8", False, 10.0 / 255, 0),
        ([-10.0, -5.0, 5.0, 10.0], "int8", True, 10.0 / 127, 0),
        ([0.0, 5.0, 10.0], "uint8", False, 10.0 / 255, 0),
        ([0.0, 5.0, 10.0], "uint8", True, 10.0 / 127, 128),
    ]
    for vals, qtype, sym, es, ez in qparam_kats:
        for mse in (False, True):
            s, z = R.utils._compute_qparams_from_array(
                np.array(vals), QT[qtype], ST["tensor"], -1, sym, False, 1.0, mse,
                np.float32, QT[qtype].np_dtype)
            np.testing.assert_allclose(s, np.float32(es), rtol=1e-5)
            assert int(z) == ez
    qrange_table = []
    for qtype in ("int4", "uint4", "int8", "uint8", "int32", "uint32"):
        for sym, red in ((False, False), (True, False), (True, True), (False, True)):
            qrange_table.append([qtype, sym, red, list(QT[qtype].qrange(sym, red))])
    pack_kats = [
        # test_pack.py:11-27, :59-75 (array, dtype, expected bytes)
        ([3, 7], "int4", [115]),
        ([-5, 3, 4, 7, 0, 3, 7, -2], "int4", [59, 116, 48, 231]),
        ([-8, 7], "int4", [120]),
        ([0, 0, 0, 0], "int4", [0, 0]),
        ([-1, -2, -3, -4], "int4", [239, 205]),
        ([1, 2, 3], "int4", [33, 3]),
        ([3, 7], "uint4", [115]),
        ([11, 3, 4, 7, 0, 3, 7, 14], "uint4", [59, 116, 48, 231]),
        ([0, 15], "uint4", [240]),
        ([15, 15, 15, 15], "uint4", [255, 255]),
        ([1, 2, 3], "uint4", [33, 3]),
    ]
    for vals, qtype, exp in pack_kats:
        a = np.array(vals, dtype=np.int8 if qtype == "int4" else np.uint8)
        got = R.pack.pack(a, QT[qtype])
        assert got.tolist() == exp, (vals, got, exp)
    # EMA known answer  test_minmax_calibrator.py:127-144
    c = R.minmax.MinMaxCalibrator(momentum=0.8)
    c.collect("t", np.array([-1.0, 2.0, 3.0]))
    c.collect("t", np.array([-0.5, 2.5, 4.0]))
    assert np.isclose(c.data["t"].min_val, -0.9) and np.isclose(c.data["t"].max_val, 3.2)
    with open(os.path.join(out, "scalar_kats.json"), "w") as f:
        json.dump(dict(qparams=qparam_kats, qrange=qrange_table, pack=pack_kats,
                       ema=dict(momentum=0.8, batches=[[-1.0, 2.0, 3.0], [-0.5, 2.5, 4.0]],
                                min=-0.9, max=3.2)), f, indent=0)
    print("scalar_kats: ok")


def gen_minmax(out):
    arrays, meta = {}, []
    rng = np.random.default_rng(21)
    for idx, (momentum, nb, shape) in enumerate([(0.0, 5, (4, 33, 17)), (0.8, 6, (3, 129)),
                                                 (0.5, 4, (2, 5, 7, 11)), (0.0, 3, (1,)),
                                                 (0.0, 4, (7, 64, 40))]):
        cal = R.minmax.MinMaxCalibrator(momentum=momentum)
        scale = np.float32(1.0)
        for b in range(nb):
            x = (rng.standard_normal(shape, dtype=np.float32) * scale
                 + np.float32(0.3 * b - 0.5)).astype(np.float32)
            scale = np.float32(scale * 1.3)
            cal.collect("t", x)
            arrays[f"s{idx}_b{b}"] = x
            arrays[f"s{idx}_b{b}_min"] = np.asarray(cal.data["t"].min_val)
            arrays[f"s{idx}_b{b}_max"] = np.asarray(cal.data["t"].max_val)
        lo, hi = cal.compute_range("t")
        arrays[f"s{idx}_lo"] = lo
        arrays[f"s{idx}_hi"] = hi
        for qtype, sym in (("int8", False), ("uint8", False), ("int8", True)):
            s, z = R.utils._compute_qparams(lo, hi, QT[qtype], sym, False,
                                            np.dtype(np.float32), QT[qtype].np_dtype)
            arrays[f"s{idx}_{qtype}_{int(sym)}_scale"] = np.asarray(s)
            arrays[f"s{idx}_{qtype}_{int(sym)}_zp"] = np.asarray(z)
        meta.append(dict(id=f"s{idx}", momentum=momentum, batches=nb, shape=list(shape)))
    np.savez_compressed(os.path.join(out, "minmax.npz"), **arrays)
    with open(os.path.join(out, "minmax.json"), "w") as f:
        json.dump(meta, f, indent=0)
    print("minmax:", len(meta), "sequences")


def gptq_call(w, x, qtype, strategy, g, sym, red, clip, block, damp, actorder, mse):
    q, s, z = R.gptq._gptq_quantize(
        w, x, quant_type=QT[qtype], strategy=ST[strategy], group_size=g, is_symmetric=sym,
        reduce_range=red, clip_ratio=clip, block_size=block, percdamp=damp,
        actorder=actorder, mse=mse, scale_dtype=np.float32, zp_dtype=QT[qtype].np_dtype)
    return np.ascontiguousarray(q), np.asarray(s), np.asarray(z)


def gen_gptq(out):
    arrays, cases = {}, []
    # (a) the reference test's own tensors: rng(42) W(16x32) then X(32x16)  test_gptq.py:8-17
    rng = np.random.default_rng(42)
    w0 = rng.normal(0, 1, (16, 32)).astype(np.float32)
    x0 = rng.normal(0, 1, (32, 16)).astype(np.float32)
    arrays["a_w"], arrays["a_x"] = w0, x0
    idx = 0
    combos = list(itertools.product(
        (("int4", "group", 8), ("int8", "tensor", 8), ("int8", "tensor", 64),
         ("uint8", "channel", -1), ("int4", "group", 16), ("uint4", "group", 4),
         ("int8", "tensor", -1), ("int8", "channel", 32)),
        (16, 128, 5), (0.01, 0.1), (False, True)))
    for (qtype, strategy, g), block, damp, actorder in combos:
        if actorder and strategy == "group":
            # group + actorder is well defined only without diag(H) ties; rng data has none
            pass
        q, s, z = gptq_call(w0, x0, qtype, strategy, g, False, False, 1.0, block, damp,
                            actorder, False)
        cid = f"a{idx:03d}"
        arrays[f"{cid}_q"], arrays[f"{cid}_s"], arrays[f"{cid}_z"] = q, s, z
        cases.append(dict(id=cid, data="a", qtype=qtype, strategy=strategy, group_size=g,
                          symmetric=False, reduce_range=False, clip_ratio=1.0,
                          block_size=block, percdamp=damp, actorder=actorder, mse=False))
        idx += 1
    # (b) a layer-shaped case with dead input channels, 3-D activations, sym / reduce_range
    rng = np.random.default_rng(43)
    k, n = 256, 96
    w1 = (rng.standard_normal((k, n), dtype=np.float32) * np.float32(0.05))
    x1 = rng.standard_normal((6, 40, k), dtype=np.float32)
    x1 *= (np.float32(0.1) + rng.random(k, dtype=np.float32) * np.float32(4.0))
    x1[..., [3, 77, 200]] = 0.0          # dead channels -> diag(H) == 0
    arrays["b_w"], arrays["b_x"] = w1, x1
    h = np.zeros((k, k), np.float32)
    h, ns = R.gptq._accumulate_hessian(x1, h, 0)
    arrays["b_h"] = h
    arrays["b_nsamples"] = np.int64(ns)
    for jdx, (qtype, strategy, g, sym, red, clip, block, damp, actorder) in enumerate([
            ("int4", "group", 128, False, False, 1.0, 128, 0.01, False),
            ("int4", "group", 128, False, False, 1.0, 128, 0.01, True),
            ("int4", "group", 32, True, False, 1.0, 64, 0.01, False),
            ("uint4", "group", 64, False, False, 0.9, 128, 0.05, False),
            ("int8", "channel", -1, False, False, 1.0, 128, 0.01, False),
            ("int8", "channel", -1, True, True, 1.0, 96, 0.01, True),
            ("uint8", "tensor", -1, False, False, 1.0, 128, 0.01, False),
            ("int8", "group", 256, False, False, 1.0, 128, 0.01, False)]):
        q, s, z = gptq_call(w1, x1, qtype, strategy, g, sym, red, clip, block, damp,
                            actorder, False)
        cid = f"b{jdx:03d}"
        arrays[f"{cid}_q"], arrays[f"{cid}_s"], arrays[f"{cid}_z"] = q, s, z
        cases.append(dict(id=cid, data="b", qtype=qtype, strategy=strategy, group_size=g,
                          symmetric=sym, reduce_range=red, clip_ratio=clip, block_size=block,
                          percdamp=damp, actorder=actorder, mse=False))
    # (c) two-step Hessian accumulate (running average algebra) on split batches
    h2 = np.zeros((k, k), np.float32)
    h2, n2 = R.gptq._accumulate_hessian(x1[:4], h2, 0)
    h2, n2 = R.gptq._accumulate_hessian(x1[4:], h2, n2)
    arrays["b_h_two_step"] = h2
    np.savez_compressed(os.path.join(out, "gptq.npz"), **arrays)
    with open(os.path.join(out, "gptq.json"), "w") as f:
        json.dump(cases, f, indent=0)
    print("gptq:", len(cases), "cases")


def gen_hqq(out):
    """core/_algorithms/hqq.py::_hqq_quantize (uint4, group, float zero points) on small matrices."""
    cases, arrays = [], {}
    grid = [
        # (kind, seed, k, n, g, reduce_range, clip, mse, lp_norm, beta, kappa, iters, early_stop)
        ("normal", 42, 32, 64, 16, False, 1.0, False, 0.7, 10.0, 1.01, 20, True),
        ("normal", 42, 32, 64, 32, False, 1.0, False, 0.7, 10.0, 1.01, 20, False),
        ("normal", 43, 64, 48, 64, False, 1.0, True, 0.7, 10.0, 1.01, 20, True),
        ("normal", 44, 128, 40, 32, False, 1.0, False, 0.5, 5.0, 1.05, 10, True),
        ("normal", 45, 128, 40, 64, False, 1.0, False, 1.0, 15.0, 1.02, 15, False),
        ("heavy", 46, 256, 36, 128, False, 1.0, False, 0.7, 10.0, 1.01, 20, True),
        ("heavy", 47, 256, 36, 128, True, 0.9, False, 0.7, 10.0, 1.01, 20, False),
        ("zero_groups", 48, 128, 24, 32, False, 1.0, False, 0.7, 10.0, 1.01, 20, True),
        ("normal", 49, 96, 20, -1, False, 1.0, False, 0.7, 10.0, 1.01, 20, True),
        ("normal", 50, 512, 16, 256, False, 1.0, False, 0.7, 10.0, 1.01, 6, False),
    ]
    for idx, (kind, seed, k, n, g, red, clip, mse, lp, beta, kappa, iters, es) in enumerate(grid):
        w = weight(kind, seed, k, n)
        q, s, z = R.hqq._hqq_quantize(w, QT["uint4"], g, reduce_range=red, clip_ratio=clip, mse=mse, lp_norm=lp, beta=beta,
                                      kappa=kappa, iters=iters, early_stop=es)
        assert s.dtype == np.float32 and z.dtype == np.float32
        key = f"c{idx}"
        arrays[key + "_q"] = np.asarray(q).astype(np.uint8)
        arrays[key + "_s"] = np.asarray(s)
        arrays[key + "_z"] = np.asarray(z)
        cases.append(dict(key=key, kind=kind, seed=seed, k=k, n=n, group_size=g, reduce_range=red, clip_ratio=clip, mse=mse,
                          lp_norm=lp, beta=beta, kappa=kappa, iters=iters, early_stop=es, w_sha=sha16(w)))
    np.savez_compressed(os.path.join(out, "hqq.npz"), **arrays)
    with open(os.path.join(out, "hqq.json"), "w") as f:
        json.dump({"cases": cases}, f, indent=1)
    print(f"hqq: {len(cases)} cases")


def _load_passes():
    """The AWQ / SmoothQuant passes (pre_passes/awq.py, smooth_quant.py) are methods on graph nodes.  Their arithmetic is
    run UNMODIFIED; what the image lacks (onnx_ir's graph objects) is replaced by the barest carriers: a tensor that
    returns its array, a value that has a name and a constant, a node / model made of namespaces.  The two methods that
    only edit the graph (`is_valid_node`, `_insert_mul_node_before`) are overridden to accept the node and to record
    the scale the pass computed."""
    ir = sys.modules["onnx_ir"]

    class Tensor:
        def __init__(self, a, dtype=None):
            self._a = np.asarray(a) if dtype is None else np.asarray(a, dtype=dtype.numpy())

        def numpy(self):
            return self._a

    class Value:
        def __init__(self, name, const_value=None):
            self.name, self.const_value = name, const_value

    ir.passes = types.SimpleNamespace(InPlacePass=object, PassResult=lambda model, modified: (model, modified))
    ir.tensor = Tensor
    ir.val = lambda name, const_value=None: Value(name, const_value)
    ir.convenience = types.SimpleNamespace(get_const_tensor=lambda v: v.const_value, replace_all_uses_with=lambda a, b: None)
    pre = types.ModuleType("onnx_quantize.pre_passes")
    pre.__path__ = [os.path.join(REF, "pre_passes")]
    sys.modules["onnx_quantize.pre_passes"] = pre
    awq = importlib.import_module("onnx_quantize.pre_passes.awq")
    sq = importlib.import_module("onnx_quantize.pre_passes.smooth_quant")

    def node_and_model(x, w, qconfig):
        node = types.SimpleNamespace(op_type="MatMul", domain="", attributes={}, meta={"qconfig": qconfig.model_dump(), "input": x.copy()},
                                     inputs=[Value("x"), Value("w", Tensor(w.copy()))], outputs=[Value("y")])
        model = types.SimpleNamespace(graph=types.SimpleNamespace(initializers={}))
        return node, model

    class Awq(awq.AwqPass):
        def is_valid_node(self, node):
            return True

        def _insert_mul_node_before(self, node, model, scale_initializer):
            self.recorded = scale_initializer.const_value.numpy()

    class Sq(sq.SmoothQuantPass):
        def _insert_mul_node_before(self, node, model, scale_initializer):
            self.recorded = scale_initializer.const_value.numpy()

    return types.SimpleNamespace(awq=awq, sq=sq, Awq=Awq, Sq=Sq, node_and_model=node_and_model)


def gen_awq(out):
    """pre_passes/awq.py::_apply_awq / _apply_awq_clip and smooth_quant.py::_smooth_quant_node on small layers."""
    P = _load_passes()
    Q = R.qconfig
    cases, arrays = [], {}
    grid = [
        # (seed, t, k, n, qtype, strategy, group, symmetric)
        (1, 48, 64, 40, "uint4", "group", 16, False),
        (2, 48, 64, 40, "uint4", "group", 32, True),
        (3, 40, 32, 24, "int8", "channel", None, False),
        (4, 40, 32, 24, "uint8", "tensor", None, False),
        (5, 64, 128, 16, "int4", "group", 64, False),
        (6, 30, 48, 20, "int8", "channel", None, True),
    ]
    for idx, (seed, t, k, n, qtype, strategy, g, sym) in enumerate(grid):
        rng = np.random.default_rng(seed)
        x = (rng.standard_normal((t, k)) * rng.uniform(0.2, 4.0, size=k)).astype(np.float32)
        w = (rng.standard_normal((k, n)) * 0.1).astype(np.float32)
        wargs = Q.QWeightArgs(dtype=QT[qtype], symmetric=sym, group_size=g, strategy=strategy)
        qcfg = Q.QConfig(weights=wargs, preprocessors=[P.awq.AwqConfig(clip_search=True)])
        key = f"c{idx}"
        node, model = P.node_and_model(x, w, qcfg)
        pas = P.Awq(clip_search=True, target_op_types={"MatMul"})
        assert pas._apply_awq(node, model)
        arrays[key + "_x"], arrays[key + "_w"] = x, w
        arrays[key + "_awq_inv_scale"] = np.asarray(pas.recorded)                       # 1 / best_scale, as emitted
        arrays[key + "_awq_w"] = model.graph.initializers["w"].const_value.numpy()      # W * best_scale
        arrays[key + "_awq_x"] = node.meta["input"]                                     # X / best_scale
        node2, _ = P.node_and_model(x, w, qcfg)
        assert pas._apply_awq_clip(node2)
        clip = float(node2.meta["qconfig"]["weights"]["clip_ratio"])
        for alpha in (0.5, 0.8):
            qs = Q.QConfig(weights=wargs, preprocessors=[P.sq.SmoothQuantConfig(alpha=alpha)])
            node3, model3 = P.node_and_model(x, w, qs)
            sp = P.Sq(alpha=alpha, target_op_types={"MatMul"})
            assert sp._smooth_quant_node(node3, model3)
            arrays[key + f"_sq{int(alpha * 10)}_inv_scale"] = np.asarray(sp.recorded)
            arrays[key + f"_sq{int(alpha * 10)}_w"] = model3.graph.initializers["w"].const_value.numpy()
        cases.append(dict(key=key, seed=seed, t=t, k=k, n=n, qtype=qtype, strategy=strategy, group_size=g, symmetric=sym, clip_ratio=clip))
    np.savez_compressed(os.path.join(out, "awq.npz"), **arrays)
    with open(os.path.join(out, "awq.json"), "w") as f:
        json.dump({"cases": cases}, f, indent=1)
    print(f"awq / smooth_quant: {len(cases)} cases")


def gen_calibrate(out):
    """core/_calibration/calibrate.py::calibrate_model -- the ORDER in which the activation list is walked (one calibrator,
    the input kind first, every tapped name collected in both walks) -- with only the onnxruntime session replaced:
    `_collect_activations` returns a prepared list of per-batch dicts.  `ml_dtypes` (one entry of a dtype table) and the
    graph objects are the barest stand-ins, as in `_load_passes`."""
    _load_passes()                                     # installs the graph-object carriers on the onnx_ir stand-in
    ir = sys.modules["onnx_ir"]
    ir.Model = ir.Node = ir.Value = object             # names in (evaluated) annotations only
    sys.modules.setdefault("ml_dtypes", types.SimpleNamespace(bfloat16=np.float16))
    cal = importlib.import_module("onnx_quantize.core._calibration.calibrate")
    base = importlib.import_module("onnx_quantize.core._calibration.base")
    Q = R.qconfig

    class Node:                                        # hashable (get_target_nodes builds a set)
        def __init__(self, name, x, y):
            self.op_type, self.name, self.meta = "MatMul", name, {}
            self.inputs = [ir.val(x), ir.val(name + "_w", ir.tensor(np.zeros((2, 2), np.float32)))]
            self.outputs = [ir.val(y)]

    def chain():
        nodes = [Node("fc1", "X", "h1"), Node("fc2", "h1", "h2"), Node("fc3", "h2", "Y")]
        return types.SimpleNamespace(graph=nodes), nodes

    cases, arrays = [], {}
    grid = [(0.0, "input"), (0.0, "output"), (0.0, "both"), (0.9, "input"), (0.9, "both"), (0.5, "both"), (0.3, "output")]
    for idx, (momentum, kinds) in enumerate(grid):
        rng = np.random.default_rng(100 + idx)
        names = ["X", "h1", "h2"] if kinds == "input" else ["h1", "h2", "Y"] if kinds == "output" else ["X", "h1", "h2", "Y"]
        acts = [{n: (rng.standard_normal((4, 6)) * (b + 1) * (1 + names.index(n))).astype(np.float32) for n in names} for b in range(5)]
        model, nodes = chain()
        qc = Q.QConfig(
            weights=Q.QWeightArgs(dtype=QT["uint8"]),
            input_activations=Q.QActivationArgs(dtype=QT["uint8"], is_static=True) if kinds != "output" else None,
            output_activations=Q.QActivationArgs(dtype=QT["int8"], symmetric=True, is_static=True) if kinds != "input" else None,
            calibration_params=base.CalibrationParams(momentum=momentum, num_samples=20, batch_size=4))
        cal._collect_activations = lambda *a, _acts=acts, **k: _acts          # the ONLY replaced step: no onnxruntime here
        cal.calibrate_model(model, qc)
        key = f"c{idx}"
        for b, act in enumerate(acts):
            for n, a in act.items():
                arrays[f"{key}_b{b}_{n}"] = a
        got = {}
        for node in nodes:
            for kind in ("input", "output"):
                if f"{kind}_scale" in node.meta:
                    nm = node.inputs[0].name if kind == "input" else node.outputs[0].name
                    arrays[f"{key}_{kind}_{nm}_scale"] = np.asarray(node.meta[f"{kind}_scale"])
                    arrays[f"{key}_{kind}_{nm}_zp"] = np.asarray(node.meta[f"{kind}_zero_point"])
                    got.setdefault(kind, []).append(nm)
        cases.append(dict(key=key, momentum=momentum, kinds=kinds, names=names, batches=len(acts), set=got))
    # the GPTQ branch: inputs of every node concatenated over the batches (calibrate.py:288-307)
    model, nodes = chain()
    rng = np.random.default_rng(7)
    acts = [{n: rng.standard_normal((3, 5, 8)).astype(np.float32) for n in ("X", "h1", "h2")} for _ in range(4)]
    qc = Q.QConfig(weights=Q.QWeightArgs(dtype=QT["uint8"], algorithm=R.gptq.GPTQConfig()))
    cal._collect_activations = lambda *a, **k: acts
    cal.calibrate_model(model, qc)
    for b, act in enumerate(acts):
        for n, a in act.items():
            arrays[f"gptq_b{b}_{n}"] = a
    for node in nodes:
        arrays[f"gptq_input_{node.inputs[0].name}"] = node.meta["input"]
    # batching rule (calibrate.py:150-172)
    data = np.arange(10 * 3, dtype=np.float32).reshape(10, 3)
    prep = []
    for bs, ns in ((2, 10), (5, 10), (10, 10), (20, 10), (3, 10), (4, 100), (4, 7)):
        o = cal._prepare_calibration_data(data, bs, ns)
        arrays[f"prep_{bs}_{ns}"] = np.asarray(o)
        prep.append([bs, ns])
    np.savez_compressed(os.path.join(out, "calibrate.npz"), **arrays)
    with open(os.path.join(out, "calibrate.json"), "w") as f:
        json.dump({"cases": cases, "gptq_batches": 4, "prepare": prep}, f, indent=1)
    print(f"calibrate: {len(cases)} walk-order cases + the GPTQ branch + {len(prep)} batching cases")


def gen_nbits(out):
    """qrules/_common.py::_prepare_for_matmul_nbits (the MatMulNBits wire format: B blob, scales, packed zero points) on
    what `_rtn_quantize` / `_hqq_quantize` return, for the uint4 / uint8 group cases MatMulNBits accepts."""
    _load_passes()
    ir = sys.modules["onnx_ir"]
    ir.Model = ir.Node = ir.Value = object
    ir.tape = types.SimpleNamespace(Tape=object)
    qr = types.ModuleType("onnx_quantize.qrules")      # bypass qrules/__init__.py (it pulls in onnxscript)
    qr.__path__ = [os.path.join(REF, "qrules")]
    sys.modules["onnx_quantize.qrules"] = qr
    common = importlib.import_module("onnx_quantize.qrules._common")
    Q = R.qconfig
    cases, arrays = [], {}
    grid = [  # (seed, k, n, qtype, g, hqq)
        (1, 64, 12, "uint4", 16, False), (2, 80, 9, "uint4", 16, False), (3, 128, 7, "uint4", 32, False), (4, 128, 10, "uint4", 128, False),
        (5, 96, 5, "uint8", 32, False), (6, 64, 6, "uint8", 64, False), (7, 96, 8, "uint4", 32, True), (8, 48, 4, "uint4", 16, True),
    ]
    for idx, (seed, k, n, qtype, g, hqq) in enumerate(grid):
        w = weight("normal", seed, k, n)
        if hqq:
            wargs = Q.QWeightArgs(dtype=QT[qtype], group_size=g, strategy="group", algorithm=R.hqq.HqqConfig())
            q, sc, zp = R.hqq._hqq_quantize(w, QT[qtype], g)
        else:
            wargs = Q.QWeightArgs(dtype=QT[qtype], group_size=g, strategy="group")
            q, sc, zp = rtn_call(w, qtype, "group", g, False, False, 1.0, False)
        qc = Q.QConfig(weights=wargs)
        assert common.is_matmul_nbits_compatible(qc)
        b, s2, pz = common._prepare_for_matmul_nbits(np.asarray(q).astype(np.uint8), np.asarray(sc), np.asarray(zp), qc)
        key = f"c{idx}"
        arrays[key + "_q"], arrays[key + "_s"], arrays[key + "_z"] = np.asarray(q).astype(np.uint8), np.asarray(sc), np.asarray(zp)
        arrays[key + "_blob"], arrays[key + "_scale"], arrays[key + "_zp"] = np.asarray(b), np.asarray(s2), np.asarray(pz)
        cases.append(dict(key=key, seed=seed, k=k, n=n, qtype=qtype, group_size=g, float_zero_points=hqq))
    # the two host-side decisions next to it: _resolve_group_size (:13-29) and is_matmul_nbits_compatible (:32-62)
    resolve = []
    for in_ch, gs in itertools.product((64, 100, 128), (None, 0, -1, 16, 32, 48, 128, 256)):
        v = ir.val("w", ir.tensor(np.zeros((in_ch, 4), np.float32)))
        resolve.append([in_ch, gs, common._resolve_group_size(v, gs)])
    compat = []
    for dt, gs, st, has_in, has_out in itertools.product(("uint4", "int4", "uint8", "int8"), (None, -1, 8, 16, 24, 32, 128),
                                                        (None, "group", "channel"), (False, True), (False, True)):
        try:
            qc = Q.QConfig(weights=Q.QWeightArgs(dtype=QT[dt], group_size=gs, strategy=st),
                           input_activations=Q.QActivationArgs(dtype=QT["uint8"], is_static=True) if has_in else None,
                           output_activations=Q.QActivationArgs(dtype=QT["uint8"], is_static=True) if has_out else None)
        except Exception:  # noqa: BLE001 -- combinations the config classes reject are not part of this table
            continue
        compat.append(dict(dtype=dt, group_size=gs, strategy=st, inputs=has_in, outputs=has_out,
                           compatible=bool(common.is_matmul_nbits_compatible(qc))))
    np.savez_compressed(os.path.join(out, "nbits.npz"), **arrays)
    with open(os.path.join(out, "nbits.json"), "w") as f:
        json.dump({"cases": cases, "resolve_group_size": resolve, "compatible": compat}, f, indent=1)
    print(f"nbits: {len(cases)} cases, {len(resolve)} group-size resolutions, {len(compat)} compatibility decisions")


def _describe(obj, fields):
    d = {}
    for f in fields:
        v = getattr(obj, f)
        if hasattr(v, "algorithm_type"):
            v = v.algorithm_type
        elif isinstance(v, enum.Enum):
            v = v.name if f == "dtype" else v.value
        elif isinstance(v, np.dtype):
            v = v.name
        d[f] = v
    return d


def _attempt(fn, fields):
    try:
        return dict(ok=True, fields=_describe(fn(), fields))
    except Exception as e:  # noqa: BLE001 -- the class of what the reference raises IS the datum
        return dict(ok=False, error=type(e).__name__)


def gen_config(out):
    """core/_qconfig.py: what QWeightArgs / QActivationArgs / QConfig accept, infer and reject, over a grid of arguments
    (the resolved fields, or the class of the exception)."""
    _load_passes()
    Q = R.qconfig
    algos = {"rtn": lambda: None, "gptq": lambda: R.gptq.GPTQConfig(), "hqq": lambda: R.hqq.HqqConfig()}
    wf = ["dtype", "symmetric", "group_size", "strategy", "scale_dtype", "zp_dtype", "reduce_range", "clip_ratio", "mse", "algorithm"]
    af = ["dtype", "symmetric", "group_size", "strategy", "scale_dtype", "zp_dtype", "reduce_range", "is_static"]
    weights, acts, confs = [], [], []
    for dt, sym, gs, st, red, al in itertools.product(list(QT), (False, True), (None, -1, 0, 16, 64), (None, "tensor", "channel", "group"),
                                                      (False, True), list(algos)):
        kw = dict(dtype=dt, symmetric=sym, group_size=gs, strategy=st, reduce_range=red)
        def make(kw=kw, al=al):
            a = algos[al]()
            return Q.QWeightArgs(**{**kw, "dtype": QT[kw["dtype"]]}, **({} if a is None else {"algorithm": a}))
        weights.append(dict(kw=kw, algorithm=al, **_attempt(make, wf)))
    for extra in (dict(clip_ratio=0.5), dict(clip_ratio=0.0), dict(clip_ratio=1.5), dict(clip_ratio=1.0, mse=True), dict(group_size=-2),
                  dict(scale_dtype="float16"), dict(scale_dtype="float32"), dict(dtype="uint4", group_size=32), dict(dtype="QInt4"), dict(strategy="GROUP", group_size=8)):
        weights.append(dict(kw=extra, algorithm="rtn", **_attempt(lambda e=extra: Q.QWeightArgs(**e), wf)))
    for dt, sym, static, st, gs in itertools.product(list(QT), (False, True), (False, True), (None, "tensor", "channel", "group"), (None, -1, 32)):
        kw = dict(dtype=dt, symmetric=sym, is_static=static, strategy=st, group_size=gs)
        acts.append(dict(kw=kw, **_attempt(lambda kw=kw: Q.QActivationArgs(**{**kw, "dtype": QT[kw["dtype"]]}), af)))
    wopts = {"none": None, "w8": dict(dtype="uint8"), "w8c": dict(dtype="int8", group_size=-1), "w4g": dict(dtype="uint4", group_size=32),
             "w8g": dict(dtype="uint8", group_size=32), "w4t": dict(dtype="int4")}
    aopts = {"none": None, "static_u8": dict(dtype="uint8", is_static=True), "dynamic_u8": dict(dtype="uint8", is_static=False),
             "static_i8": dict(dtype="int8", is_static=True, symmetric=True)}
    for wk, ik, ok, fmt in itertools.product(wopts, aopts, aopts, (None, "qdq", "qlinear")):
        def make(wk=wk, ik=ik, ok=ok, fmt=fmt):
            kw = {}
            if wopts[wk] is not None:
                kw["weights"] = Q.QWeightArgs(**wopts[wk])
            if aopts[ik] is not None:
                kw["input_activations"] = Q.QActivationArgs(**aopts[ik])
            if aopts[ok] is not None:
                kw["output_activations"] = Q.QActivationArgs(**aopts[ok])
            if fmt is not None:
                kw["format"] = fmt
            return Q.QConfig(**kw)
        confs.append(dict(weights=wk, inputs=ik, outputs=ok, format=fmt, **_attempt(make, ["format"])))
    # the small parameter models: algorithm / pre-pass / calibration settings
    base = importlib.import_module("onnx_quantize.core._calibration.base")
    P = sys.modules["onnx_quantize.pre_passes.awq"], sys.modules["onnx_quantize.pre_passes.smooth_quant"]
    params = []
    def add(model, cls, fields, grid):
        for kw in grid:
            params.append(dict(model=model, kw=kw, **_attempt(lambda kw=kw: cls(**kw), fields)))
    add("gptq", R.gptq.GPTQConfig, ["algorithm_type", "block_size", "percdamp", "actorder"],
        [dict()] + [dict(block_size=b) for b in (-1, 0, 1, 64, "128", 1.5)] + [dict(percdamp=p) for p in (-0.1, 0, 0.5, 1, 2, "0.1")] +
        [dict(actorder=a) for a in (True, 1, "yes", None)] + [dict(unknown=1), dict(algorithm_type="rtn")])
    add("hqq", R.hqq.HqqConfig, ["algorithm_type", "lp_norm", "beta", "kappa", "iters", "early_stop"],
        [dict()] + [dict(lp_norm=v) for v in (0, 0.5, 1, 2, -1)] + [dict(beta=v) for v in (0, -1, 5)] + [dict(kappa=v) for v in (1, 0.5, 2)] +
        [dict(iters=v) for v in (0, -1, 5, 2.5)] + [dict(early_stop=False), dict(nope=1)])
    add("awq", P[0].AwqConfig, ["preprocessing_type", "clip_search"], [dict(), dict(clip_search=True), dict(clip_search="no"), dict(x=1)])
    add("smooth_quant", P[1].SmoothQuantConfig, ["preprocessing_type", "alpha"], [dict()] + [dict(alpha=a) for a in (-0.1, 0, 0.5, 1, 1.1, "0.3")])
    add("calibration", base.CalibrationParams, ["method", "num_samples", "batch_size", "momentum", "provider"],
        [dict()] + [dict(momentum=m) for m in (-0.1, 0, 0.5, 0.99, 1, 1.2)] + [dict(num_samples=n) for n in (0, 1, -5, 7)] +
        [dict(batch_size=b) for b in (0, 1, -1)] + [dict(method=m) for m in ("minmax", "MinMax", "entropy", 3)] +
        [dict(provider=p) for p in ("cpu", "CPU", "cuda", "gpu", "rocm", "CPUExecutionProvider", "CUDAExecutionProvider", "tpu", "")] + [dict(extra=1)])
    with open(os.path.join(out, "config.json"), "w") as f:
        json.dump(dict(weight_options=wopts, activation_options=aopts, weights=weights, activations=acts, configs=confs, params=params), f)
    print(f"config: {len(params)} parameter-model cases; {len(weights)} QWeightArgs, {len(acts)} QActivationArgs, {len(confs)} QConfig combinations "
          f"({sum(not w['ok'] for w in weights)} / {sum(not a['ok'] for a in acts)} / {sum(not c['ok'] for c in confs)} rejected)")


def gen_seam(out):
    """The plugin seam itself: `qconfig.weights.algorithm.quantize_weights(w, qconfig, out=out)` (qrules/_common.py:133) of the
    reference's RTNConfig / GPTQConfig / HqqConfig, called with carrier values (a weight value whose constant returns its
    array, an output value whose producer node carries the calibration inputs in `meta["input"]`)."""
    _load_passes()
    ir = sys.modules["onnx_ir"]
    Q = R.qconfig
    cases, arrays = [], {}
    grid = [  # (algorithm, seed, k, n, kwargs of QWeightArgs, kwargs of the algorithm config)
        ("rtn", 1, 64, 48, dict(dtype="int8", symmetric=True), {}),
        ("rtn", 2, 96, 40, dict(dtype="uint8", group_size=-1), {}),
        ("rtn", 3, 128, 24, dict(dtype="uint4", group_size=32), {}),
        ("rtn", 4, 128, 24, dict(dtype="int4", group_size=64, symmetric=True, clip_ratio=0.9), {}),
        ("gptq", 5, 64, 32, dict(dtype="int4", group_size=32), dict(block_size=32)),
        ("gptq", 6, 96, 20, dict(dtype="int8", group_size=-1), dict(percdamp=0.05)),
        ("gptq", 7, 64, 16, dict(dtype="uint8"), dict(actorder=True)),
        ("hqq", 8, 128, 20, dict(dtype="uint4", group_size=32, strategy="group"), dict(iters=10)),
    ]
    algos = {"rtn": None, "gptq": R.gptq.GPTQConfig, "hqq": R.hqq.HqqConfig}
    for idx, (al, seed, k, n, wkw, akw) in enumerate(grid):
        w = weight("normal", seed, k, n) * np.float32(0.1)
        rng = np.random.default_rng(seed + 100)
        x = (rng.standard_normal((6, 10, k)) * rng.uniform(0.3, 3.0, size=k)).astype(np.float32)
        kw = {**wkw, "dtype": QT[wkw["dtype"]]}
        if algos[al] is not None:
            kw["algorithm"] = algos[al](**akw)
        qc = Q.QConfig(weights=Q.QWeightArgs(**kw))
        node = types.SimpleNamespace(meta={"input": x})
        outv = types.SimpleNamespace(producer=lambda node=node: node)
        q, sc, zp = qc.weights.algorithm.quantize_weights(ir.val("w", ir.tensor(w)), qc, out=outv)
        key = f"c{idx}"
        arrays[key + "_w"], arrays[key + "_x"] = w, x
        arrays[key + "_q"] = np.asarray(q).astype(np.int8 if wkw["dtype"].startswith("int") else np.uint8)
        arrays[key + "_s"], arrays[key + "_z"] = np.asarray(sc), np.asarray(zp).astype(np.float32 if al == "hqq" else np.int32)
        cases.append(dict(key=key, algorithm=al, weights=wkw, config=akw, q_shape=list(np.shape(q)), s_shape=list(np.shape(sc)),
                          z_shape=list(np.shape(zp)), s_dtype=str(np.asarray(sc).dtype), z_dtype=str(np.asarray(zp).dtype)))
    np.savez_compressed(os.path.join(out, "seam.npz"), **arrays)
    with open(os.path.join(out, "seam.json"), "w") as f:
        json.dump({"cases": cases}, f, indent=1)
    print(f"seam: {len(cases)} cases")


def _load_common():
    """qrules/_common.py with the carriers of `_load_passes` (its package __init__ pulls in onnxscript and is bypassed)."""
    _load_passes()
    ir = sys.modules["onnx_ir"]
    ir.Model = ir.Node = ir.Value = object
    ir.tape = types.SimpleNamespace(Tape=object)
    if "onnx_quantize.qrules" not in sys.modules:
        qr = types.ModuleType("onnx_quantize.qrules")
        qr.__path__ = [os.path.join(REF, "qrules")]
        sys.modules["onnx_quantize.qrules"] = qr
    return importlib.import_module("onnx_quantize.qrules._common")


class RecordingTape:
    """The `op` a rewrite rule receives, reduced to what the seam touches: `initializer(tensor, name=)` records the array."""

    def __init__(self):
        self.initializers = []

    def initializer(self, tensor, name=None):
        ir = sys.modules["onnx_ir"]
        self.initializers.append((name, np.asarray(tensor.numpy())))
        return ir.val(name, tensor)


def gen_seam_qw(out):
    """The ONE function every rewrite rule calls: qrules/_common.py::quantize_weights(op, w, qconfig, out,
    is_matmul_nbits_compatible) -- algorithm plugin + `_prepare_for_matmul_nbits` + the three `op.initializer` calls --
    run unmodified on a recording tape; `is_matmul_nbits_compatible` is decided by the reference's own predicate and
    the group size resolved by its `_resolve_group_size`, as `QRewriter._rewrite` / `_rewrite_weights_only` do
    (qrules/base.py:72, _qdq/matmul_to_qmatmul.py:84-88)."""
    common = _load_common()
    ir = sys.modules["onnx_ir"]
    Q = R.qconfig
    cases, arrays = [], {}
    grid = [  # (algorithm, seed, k, n, kwargs of QWeightArgs, kwargs of the algorithm config)
        ("rtn", 1, 128, 24, dict(dtype="uint4", group_size=32), {}),
        ("rtn", 2, 80, 9, dict(dtype="uint4", group_size=16), {}),                      # odd number of blocks: 0x8 pad nibble
        ("rtn", 3, 128, 10, dict(dtype="uint4", group_size=128), {}),                   # one block: zero points not packed
        ("rtn", 4, 128, 6, dict(dtype="uint8", group_size=64), {}),
        ("rtn", 5, 96, 8, dict(dtype="uint4", group_size=256), {}),                     # resolved to K = 96: not a power of two, still flagged by the caller's predicate on the ORIGINAL size
        ("rtn", 6, 100, 12, dict(dtype="uint4", group_size=32), {}),                    # 32 does not divide 100 -> K
        ("rtn", 7, 64, 16, dict(dtype="uint4", group_size=-1), {}),                     # -1 passes the predicate and breaks the packer
        ("rtn", 8, 128, 24, dict(dtype="int4", group_size=64, symmetric=True, clip_ratio=0.9), {}),
        ("rtn", 9, 96, 40, dict(dtype="int8", strategy="channel"), {}),
        ("rtn", 10, 64, 48, dict(dtype="uint8"), {}),
        ("rtn", 11, 256, 16, dict(dtype="uint4", group_size=64, symmetric=True), {}),
        ("gptq", 12, 64, 32, dict(dtype="uint4", group_size=32), dict(block_size=32)),
        ("gptq", 13, 96, 20, dict(dtype="int4", group_size=32), {}),
        ("hqq", 14, 128, 20, dict(dtype="uint4", group_size=32, strategy="group"), dict(iters=10)),
        ("hqq", 15, 64, 12, dict(dtype="uint4", group_size=64, strategy="group"), {}),
        ("rtn", 17, 64, 8, dict(dtype="uint4", group_size=128), {}),                     # oversize group resolved to K = 64, a power of two
        ("rtn", 18, 64, 8, dict(dtype="uint8", group_size=16, reduce_range=True), {}),
    ]
    algos = {"rtn": None, "gptq": R.gptq.GPTQConfig, "hqq": R.hqq.HqqConfig}
    for idx, (al, seed, k, n, wkw, akw) in enumerate(grid):
        w = weight("normal", seed, k, n) * np.float32(0.1)
        rng = np.random.default_rng(seed + 100)
        x = (rng.standard_normal((6, 10, k)) * rng.uniform(0.3, 3.0, size=k)).astype(np.float32)
        kw = {**wkw, "dtype": QT[wkw["dtype"]]}
        if algos[al] is not None:
            kw["algorithm"] = algos[al](**akw)
        qc = Q.QConfig(weights=Q.QWeightArgs(**kw))
        wv = ir.val("fc.weight", ir.tensor(w))
        qc.weights.group_size = common._resolve_group_size(wv, qc.weights.group_size)        # qrules/base.py:72
        flagged = bool(common.is_matmul_nbits_compatible(qc, wv.name))
        node = types.SimpleNamespace(meta={"input": x})
        outv = types.SimpleNamespace(producer=lambda node=node: node)
        tape = RecordingTape()
        key = f"c{idx}"
        arrays[key + "_w"], arrays[key + "_x"] = w, x
        case = dict(key=key, algorithm=al, weights=wkw, config=akw, k=k, n=n, resolved_group_size=qc.weights.group_size, flagged=flagged)
        try:
            common.quantize_weights(tape, wv, qc, outv, is_matmul_nbits_compatible=flagged)
        except Exception as e:  # noqa: BLE001 -- the exception class is the recorded behaviour
            case["raises"] = type(e).__name__
            cases.append(case)
            continue
        case["initializers"] = []
        for j, (name, a) in enumerate(tape.initializers):
            store = a.astype(np.float32) if a.dtype.kind == "f" else a.astype(np.int32)
            arrays[f"{key}_i{j}"] = store
            case["initializers"].append(dict(name=name, shape=list(a.shape), dtype=str(a.dtype)))
        cases.append(case)
    np.savez_compressed(os.path.join(out, "seam_qw.npz"), **arrays)
    with open(os.path.join(out, "seam_qw.json"), "w") as f:
        json.dump({"cases": cases}, f, indent=1)
    print(f"seam_qw: {len(cases)} cases ({sum('raises' in c for c in cases)} raising)")


def _load_rules():
    """The rewrite rules (qrules/_qdq/*.py, qrules/_qlinear/*.py, qrules/base.py) and the function-name factories
    (qfunctions/factory.py) need `onnxscript` only for (a) the base class of a rule, whose `rule()` is called once at
    import, (b) `MatchResult` inside `check` (not run here), (c) the `@script` decorator of the emitted functions, whose
    bodies are never executed at quantize time.  Those three names are stood in for; every `_rewrite*` method, the
    `_common` helpers and the factories run unmodified."""
    common = _load_common()
    ir = sys.modules["onnx_ir"]
    if "onnxscript" not in sys.modules:
        osc = types.ModuleType("onnxscript")

        class RewriteRuleClassBase:
            def rule(self):
                return self

        class Opset:
            def __init__(self, domain, version):
                self.domain, self.version = domain, version

        def script(opset=None, **_):
            def deco(fn):
                fn.to_function_proto = lambda: fn.__name__
                return fn
            return deco

        osc.rewriter = types.SimpleNamespace(RewriteRuleClassBase=RewriteRuleClassBase, MatchResult=object)
        osc.values = types.SimpleNamespace(Opset=Opset)
        osc.script = script
        osc.opset21 = types.SimpleNamespace()
        sys.modules["onnxscript"] = osc
        ir.AttrInt64 = lambda name, value: types.SimpleNamespace(name=name, value=value)
        qf = types.ModuleType("onnx_quantize.qfunctions")        # its __init__ deserialises function protos: bypassed
        qf.__path__ = [os.path.join(REF, "qfunctions")]
        sys.modules["onnx_quantize.qfunctions"] = qf
        reg = importlib.import_module("onnx_quantize.qfunctions.register")
        fac = importlib.import_module("onnx_quantize.qfunctions.factory")
        qf.MS_OPSET, qf.QUANT_OPSET, qf.get_qfunction = reg.MS_OPSET, reg.QUANT_OPSET, fac.get_qfunction
        qf.factory = fac
    mods = types.SimpleNamespace(common=common)
    mods.qdq_matmul = importlib.import_module("onnx_quantize.qrules._qdq.matmul_to_qmatmul")
    mods.qdq_gemm = importlib.import_module("onnx_quantize.qrules._qdq.gemm_to_qgemm")
    mods.ql_matmul = importlib.import_module("onnx_quantize.qrules._qlinear.matmul_to_qmatmul")
    mods.ql_gemm = importlib.import_module("onnx_quantize.qrules._qlinear.gemm_to_qgemm")
    return mods


class RecordingOp(RecordingTape):
    """The rewriter's `op`: initializers are recorded, any other attribute is an operator / function call whose name,
    input value names and attributes are recorded."""

    def __init__(self):
        super().__init__()
        self.calls = []

    def __getattr__(self, name):
        ir = sys.modules["onnx_ir"]

        def call(*args, **kw):
            self.calls.append((name, [None if a is None else a.name for a in args], dict(kw)))
            return ir.val(f"{name}/out")
        return call


def gen_emit(out):
    """The emission contract without the ONNX stack: every rule class's `_rewrite` (qrules/base.py:51-81
    dispatch -> `_rewrite_weights_only[_standard|_matmul_nbits]` / `_rewrite_static` / `_rewrite_dynamic`,
    `_get_activation_qparams` :15-40) run on a recording `op` for the rule paths of the five BASELINE configurations and
    the Gemm / QLinear variants next to them: initializer names, shapes, dtypes and values, the emitted function or
    operator name, its input order, attributes, domain and version."""
    M = _load_rules()
    ir = sys.modules["onnx_ir"]
    Q = R.qconfig
    act = lambda dt, **kw: Q.QActivationArgs(dtype=QT[dt], **kw)      # noqa: E731
    grid = [  # (id, rule class, has bias, k, n, QWeightArgs kwargs, algorithm, QConfig kwargs)
        ("config1_int8_sym_tensor", M.qdq_matmul.MatMulToQMatMul, False, 256, 512, dict(dtype="int8", symmetric=True), None, {}),
        ("config2_uint4_g128_nbits", M.qdq_matmul.MatMulToQMatMul, False, 256, 24, dict(dtype="uint4", group_size=128), None, {}),
        ("config3_static_qdq", M.qdq_matmul.MatMulToQMatMul, False, 64, 48, dict(dtype="int8"), None,
         dict(input_activations=act("int8", is_static=True), output_activations=act("int8", is_static=True))),
        ("config3_static_qlinear", M.ql_matmul.MatMulToQLinearMatMul, False, 64, 48, dict(dtype="int8", symmetric=True), None,
         dict(format="qlinear", input_activations=act("uint8", is_static=True), output_activations=act("uint8", is_static=True))),
        ("config4_gptq_int4_g128", M.qdq_matmul.MatMulToQMatMul, False, 256, 16, dict(dtype="int4", group_size=128), "gptq", {}),
        ("matmul_int8_group32", M.qdq_matmul.MatMulToQMatMul, False, 64, 12, dict(dtype="int8", group_size=32), None, {}),
        ("matmul_dynamic_input", M.qdq_matmul.MatMulToQMatMul, False, 64, 12, dict(dtype="int8", strategy="channel"), None,
         dict(input_activations=act("uint8", is_static=False))),
        ("matmul_static_input_only", M.qdq_matmul.MatMulToQMatMul, False, 64, 12, dict(dtype="uint8"), None,
         dict(input_activations=act("uint8", is_static=True))),
        ("gemm_bias_uint4_g32_nbits", M.qdq_gemm.GemmBiasToQGemmBias, True, 64, 12, dict(dtype="uint4", group_size=32), None, {}),
        ("gemm_bias_int4_g32_grouped", M.qdq_gemm.GemmBiasToQGemmBias, True, 64, 12, dict(dtype="int4", group_size=32), None, {}),
        ("gemm_bias_static_qdq", M.qdq_gemm.GemmBiasToQGemmBias, True, 64, 12, dict(dtype="int8", symmetric=True, strategy="channel"), None,
         dict(input_activations=act("int8", is_static=True), output_activations=act("int8", is_static=True))),
        ("gemm_bias_dynamic_qdq", M.qdq_gemm.GemmBiasToQGemmBias, True, 64, 12, dict(dtype="int8"), None,
         dict(input_activations=act("uint8", is_static=False))),
        ("gemm_nobias_int8_channel", M.qdq_gemm.GemmToQGemm, False, 64, 12, dict(dtype="int8", strategy="channel"), None, {}),
        ("gemm_bias_static_qlinear", M.ql_gemm.GemmBiasToQLinearGemmBias, True, 64, 12, dict(dtype="int8", symmetric=True), None,
         dict(format="qlinear", input_activations=act("uint8", is_static=True), output_activations=act("uint8", is_static=True))),
    ]
    cases, arrays = [], {}
    for idx, (cid, rule_cls, has_bias, k, n, wkw, algo, ckw) in enumerate(grid):
        rng = np.random.default_rng(500 + idx)
        w = (rng.standard_normal((k, n)) * 0.1).astype(np.float32)
        b = (rng.standard_normal(n) * 0.05).astype(np.float32)
        x = (rng.standard_normal((4, 8, k)) * rng.uniform(0.3, 3.0, size=k)).astype(np.float32)
        kw = {**wkw, "dtype": QT[wkw["dtype"]]}
        if algo == "gptq":
            kw["algorithm"] = R.gptq.GPTQConfig()
        qc = Q.QConfig(weights=Q.QWeightArgs(**kw), **ckw)
        meta = {"qconfig": qc.model_dump(), "input": x}
        for kind, aargs in (("input", qc.input_activations), ("output", qc.output_activations)):
            if aargs is not None and aargs.is_static:
                meta[f"{kind}_scale"] = np.array(0.02 + 0.01 * idx + (0.005 if kind == "output" else 0), dtype=np.float32)
                meta[f"{kind}_zero_point"] = np.array(3 if kind == "input" else 5).astype(aargs.dtype.np_dtype)
        node = types.SimpleNamespace(meta=meta, outputs=[ir.val("fc/out")], attributes={})
        outv = ir.val("fc/out")
        outv.producer = lambda node=node: node
        op = RecordingOp()
        args = [ir.val("X"), ir.val("fc.weight", ir.tensor(w))] + ([ir.val("fc.bias", ir.tensor(b))] if has_bias else []) + [outv]
        rule = rule_cls()
        rule._rewrite(op, *args)
        key = f"e{idx}"
        arrays[key + "_w"], arrays[key + "_b"], arrays[key + "_x"] = w, b, x
        inits = []
        for j, (name, a) in enumerate(op.initializers):
            arrays[f"{key}_i{j}"] = a.astype(np.float32) if a.dtype.kind == "f" else a.astype(np.int64)
            inits.append(dict(name=name, shape=list(a.shape), dtype=str(a.dtype)))
        assert len(op.calls) == 1
        cname, cin, ckw2 = op.calls[0]
        cases.append(dict(id=cid, key=key, rule=rule_cls.__name__, op_type=rule.op_type, has_bias=has_bias, k=k, n=n, weights=wkw, algorithm=algo,
                          format=qc.format.value,
                          input_activations=None if qc.input_activations is None else dict(dtype=_describe(qc.input_activations, ["dtype"])["dtype"], is_static=qc.input_activations.is_static),
                          output_activations=None if qc.output_activations is None else dict(dtype=_describe(qc.output_activations, ["dtype"])["dtype"], is_static=qc.output_activations.is_static),
                          meta={m: (float(v) if m.endswith("scale") else int(v)) for m, v in meta.items() if m not in ("qconfig", "input")},
                          initializers=inits, call=dict(name=cname, inputs=cin, attrs={a: v for a, v in ckw2.items()})))
    np.savez_compressed(os.path.join(out, "emit.npz"), **arrays)
    with open(os.path.join(out, "emit.json"), "w") as f:
        json.dump({"cases": cases}, f, indent=1)
    print(f"emit: {len(cases)} rule paths")


def gen_digests(out):
    """Digests of the BASELINE.json configurations (inputs are regenerated from seeds)."""
    d = {}
    w = weight("normal", 0, 256, 512)
    q, s, z = rtn_call(w, "int8", "tensor", -1, True, False, 1.0, False)
    d["config1"] = dict(k=256, n=512, seed=0, kind="normal", qtype="int8", strategy="tensor",
                        group_size=-1, symmetric=True, w_sha=sha16(w), q_sha=sha16(q),
                        scale_hex=s.tobytes().hex(), zp=int(z))
    w = weight("normal", 0, 4096, 11008)
    for sym in (False, True):
        q, s, z = rtn_call(w, "uint4", "group", 128, sym, False, 1.0, False)
        d[f"config2_{'sym' if sym else 'asym'}"] = dict(
            k=4096, n=11008, seed=0, kind="normal", qtype="uint4", strategy="group",
            group_size=128, symmetric=sym, w_sha=sha16(w), q_sha=sha16(q), s_sha=sha16(s),
            z_sha=sha16(z), scale0_hex=s[0].tobytes().hex(), zp_head=z[:4, 0].tolist(),
            q_head=q[:4, 0].tolist())
    w = weight("heavy", 1, 4096, 11008)
    q, s, z = rtn_call(w, "uint4", "group", 128, False, False, 1.0, False)
    d["config2_heavy"] = dict(k=4096, n=11008, seed=1, kind="heavy", qtype="uint4",
                              strategy="group", group_size=128, symmetric=False,
                              w_sha=sha16(w), q_sha=sha16(q), s_sha=sha16(s), z_sha=sha16(z))
    w = weight("zero_groups", 2, 4096, 11008)
    q, s, z = rtn_call(w, "uint4", "group", 128, False, False, 1.0, False)
    d["config2_zero_groups"] = dict(k=4096, n=11008, seed=2, kind="zero_groups", qtype="uint4",
                                    strategy="group", group_size=128, symmetric=False,
                                    w_sha=sha16(w), q_sha=sha16(q), s_sha=sha16(s),
                                    z_sha=sha16(z))
    w = weight("normal", 5, 4096, 4096)
    q, s, z = rtn_call(w, "int8", "channel", -1, False, False, 1.0, False)
    d["channel_4096"] = dict(k=4096, n=4096, seed=5, kind="normal", qtype="int8",
                             strategy="channel", group_size=-1, symmetric=False,
                             w_sha=sha16(w), q_sha=sha16(q), s_sha=sha16(s), z_sha=sha16(z))
    q, s, z = rtn_call(w, "int4", "group", 128, False, False, 1.0, False)
    d["int4_g128_4096"] = dict(k=4096, n=4096, seed=5, kind="normal", qtype="int4",
                               strategy="group", group_size=128, symmetric=False,
                               w_sha=sha16(w), q_sha=sha16(q), s_sha=sha16(s), z_sha=sha16(z))
    # round 4: the reference's DEFAULT strategies on the headline matrix (bench.py `strategies`), and the packed int4 [K, N]
    # serialisation (core/_pack.py:8-22 on what _rtn_quantize returns) of the int4 g128 configurations (configs 4 / 5)
    w = weight("normal", 0, 4096, 11008)
    for strategy in ("channel", "tensor"):
        q, s, z = rtn_call(w, "int8", strategy, -1, False, False, 1.0, False)
        d[f"headline_int8_{strategy}"] = dict(k=4096, n=11008, seed=0, kind="normal", qtype="int8", strategy=strategy, group_size=-1,
                                              symmetric=False, w_sha=sha16(w), q_sha=sha16(q), s_sha=sha16(np.asarray(s)),
                                              z_sha=sha16(np.asarray(z)))
    for qtype in ("int4", "uint4"):
        q, s, z = rtn_call(w, qtype, "group", 128, False, False, 1.0, False)
        packed = R.pack.pack(q, QT[qtype])
        d[f"headline_{qtype}_g128_packed"] = dict(k=4096, n=11008, seed=0, kind="normal", qtype=qtype, strategy="group", group_size=128,
                                                  symmetric=False, w_sha=sha16(w), q_sha=sha16(q), packed_sha=sha16(packed),
                                                  packed_bytes=int(packed.size), s_sha=sha16(s), z_sha=sha16(z))
    # Llama's down_proj shape: columns of 11008 rows (the streamed per-channel kernel; a per-tensor range over 2752 tiles)
    w = weight("normal", 6, 11008, 4096)
    for strategy in ("channel", "tensor"):
        q, s, z = rtn_call(w, "int8", strategy, -1, False, False, 1.0, False)
        d[f"tall_int8_{strategy}"] = dict(k=11008, n=4096, seed=6, kind="normal", qtype="int8", strategy=strategy, group_size=-1,
                                          symmetric=False, w_sha=sha16(w), q_sha=sha16(q), s_sha=sha16(np.asarray(s)),
                                          z_sha=sha16(np.asarray(z)))
    with open(os.path.join(out, "digests.json"), "w") as f:
        json.dump(d, f, indent=1)
    print("digests: ok")


def main():
    out = HERE
    gens = dict(scalar_kats=gen_scalar_kats, rtn_small=gen_rtn_small, rtn_mse=gen_rtn_mse, kernels=gen_kernels,
                minmax=gen_minmax, gptq=gen_gptq, hqq=gen_hqq, awq=gen_awq, calibrate=gen_calibrate, nbits=gen_nbits, config=gen_config, seam=gen_seam, seam_qw=gen_seam_qw, emit=gen_emit, digests=gen_digests)
    for name in (sys.argv[1:] or list(gens)):     # python make_golden.py [hqq ...] regenerates only the named sets
        gens[name](out)
    meta = dict(numpy=np.__version__, python=sys.version.split()[0],
                reference="/root/reference (AyoubMDL/onnx_quantize v0.3.0 checkout)",
                stand_ins=["onnx_ir: DataType enum; for the pass methods, calibrate_model, the wire-format helpers and the plugin-seam "
                           "call also Tensor (returns its array), Value (name + constant), passes.InPlacePass = object, "
                           "convenience.get_const_tensor / replace_all_uses_with, tape.Tape / Model / Node / Value names for annotations",
                           "ml_dtypes: bfloat16 entry of calibrate.py's dtype table",
                           "package objects with __path__ for onnx_quantize, onnx_quantize.pre_passes, onnx_quantize.qrules, "
                           "onnx_quantize.qfunctions (their __init__ pull in onnx / onnxscript)",
                           "onnxscript (emit set only): rewriter.RewriteRuleClassBase (a base class with rule()), rewriter.MatchResult, "
                           "values.Opset (domain + version), script (a decorator that returns the function: the emitted functions' "
                           "bodies are never executed at quantize time), opset21; onnx_ir.AttrInt64 name",
                           "recording tapes for the rewriter's `op`: initializer(tensor, name=) and operator / function calls are "
                           "recorded (seam_qw and emit sets)"],
                replaced_or_overridden=["calibrate._collect_activations (the onnxruntime session) -> a prepared list of per-batch dicts",
                                        "AwqPass.is_valid_node -> True; AwqPass / SmoothQuantPass._insert_mul_node_before -> records the "
                                        "scale initializer (graph edits only)"],
                note="every arithmetic statement of the reference runs unmodified; only data leaves the script")
    with open(os.path.join(out, "PROVENANCE.json"), "w") as f:
        json.dump(meta, f, indent=1)


if __name__ == "__main__":
    main()
