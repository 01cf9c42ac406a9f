"""The NumPy-facing API (same names and call shapes as the reference) running on the GPU: these tests
read like the reference's test/core/algorithms/test_rtn.py, test_gptq.py and
test/core/calibration/test_minmax_calibrator.py, plus golden-vector checks of the small kernels."""
import math

import numpy as np
import pytest

import oq_oracle as O
from conftest import load_json, load_npz
from onnx_quantize_amd import QuantizationStrategy, QuantType

pytestmark = pytest.mark.gpu

QT = {"int4": QuantType.QInt4, "uint4": QuantType.QUInt4, "int8": QuantType.QInt8, "uint8": QuantType.QUInt8}
KATS = load_json("scalar_kats.json")


@pytest.fixture(scope="module")
def F():
    from onnx_quantize_amd.algorithms import functional
    return functional


@pytest.fixture(scope="module")
def A():
    from onnx_quantize_amd import algorithms
    return algorithms


# ------------------------------------------------------------------ test_rtn.py:21-72
@pytest.mark.parametrize("vals,qtype,sym,exp_scale,exp_zp", KATS["qparams"])
def test_get_quantization_params_scalar(F, vals, qtype, sym, exp_scale, exp_zp):
    quant_type = QT[qtype]
    scale, zero_point = F._compute_qparams_from_array(
        np.array(vals), quant_type, QuantizationStrategy.TENSOR, group_size=-1, is_symmetric=sym,
        reduce_range=False, clip_ratio=1.0, mse=False, scale_dtype=np.float32, zp_dtype=quant_type.np_dtype)
    assert scale > 0 and scale.size == 1
    np.testing.assert_allclose(scale, np.array(exp_scale, dtype=np.float32), rtol=1e-5)
    assert zero_point.dtype == quant_type.np_dtype and zero_point.size == 1
    assert int(zero_point) == exp_zp
    qmin, qmax = quant_type.qrange(sym)
    assert qmin <= zero_point <= qmax


@pytest.mark.parametrize("fp_array, qtype, symmetric", [
    (np.array([[-5.0, 0.0, 10.0], [-2.0, 5.0, 3.0]]), "int8", False),
    (np.array([[0.0, 5.0, 10.0], [1.0, 2.0, 3.0]]), "uint8", False),
    (np.array([[-10.0, -5.0, 5.0], [2.0, 1.0, -1.0]]), "int8", True),
    (np.array([[0.0, 0.0, 0.0], [1.0, 2.0, 3.0]]), "int8", False),
])
def test_get_quantization_params_per_channel(F, fp_array, qtype, symmetric):
    quant_type = QT[qtype]
    scale, zero_point = F._compute_qparams_from_array(
        fp_array, quant_type, QuantizationStrategy.CHANNEL, group_size=-1, is_symmetric=symmetric,
        reduce_range=False, clip_ratio=1.0, mse=False, scale_dtype=np.float32, zp_dtype=quant_type.np_dtype)
    assert scale.shape == (fp_array.shape[0], 1) and zero_point.shape == (fp_array.shape[0], 1)
    assert np.all(scale > 0) and zero_point.dtype == quant_type.np_dtype
    es, ez = O.qparams_from_rows(fp_array.astype(np.float32), qtype, "channel", symmetric, False)
    np.testing.assert_allclose(scale, es, rtol=1e-6)
    np.testing.assert_array_equal(zero_point, ez)


@pytest.mark.parametrize("qtype, symmetric, group_size", [("int8", False, 2), ("uint8", False, 4), ("int8", True, 16),
                                                          ("int8", False, 7)])
def test_get_quantization_params_group(F, qtype, symmetric, group_size):
    quant_type = QT[qtype]
    fp_array = np.ones((32, 64), dtype=np.float32)
    in_channels, out_channels = fp_array.shape
    if group_size > in_channels or in_channels % group_size:       # qrules/_common.py:13-29
        group_size = in_channels
    rows = F._preprocess_array(fp_array, QuantizationStrategy.GROUP, group_size)
    scale, zero_point = F._compute_qparams_from_array(
        rows, quant_type, QuantizationStrategy.GROUP, group_size=group_size, is_symmetric=symmetric,
        reduce_range=False, clip_ratio=1.0, mse=False, scale_dtype=np.float32, zp_dtype=quant_type.np_dtype)
    num_groups = math.ceil(in_channels / group_size)
    assert scale.shape == (out_channels * num_groups, 1) == zero_point.shape
    assert np.all(scale > 0) and zero_point.dtype == quant_type.np_dtype


def test_quantize_bias(A, rng):
    bias = rng.random((16,)).astype(np.float32)
    input_scale = 1.5
    weight_scale = rng.random((16,)).astype(np.float32)
    q_bias, scale, zero_point = A._quantize_bias(bias, input_scale, weight_scale)
    assert q_bias.shape == bias.shape and q_bias.dtype == np.int32 and zero_point == 0
    np.testing.assert_array_equal(scale, input_scale * weight_scale)
    K = load_npz("kernels.npz")
    qb, bs, _ = A._quantize_bias(K["bias"], K["bias_xscale"], K["bias_wscale"])
    np.testing.assert_array_equal(qb, K["bias_q"])
    assert bs.tobytes() == K["bias_scale"].tobytes()


@pytest.mark.parametrize("strategy,group_size", [(QuantizationStrategy.TENSOR, None), (QuantizationStrategy.CHANNEL, -1),
                                                 (QuantizationStrategy.GROUP, 16), (QuantizationStrategy.GROUP, 64)])
@pytest.mark.parametrize("qtype", ["int4", "uint4", "int8", "uint8"])
@pytest.mark.parametrize("symmetric", [False, True])
def test_rtn_quantize_shapes_dtypes_error_bound(A, F, rng, strategy, group_size, qtype, symmetric):
    """test_rtn.py:259-452: shapes, dtypes, range and |dq - w| <= 2 * scale."""
    quant_type = QT[qtype]
    w = rng.normal(0, 1, (64, 48)).astype(np.float32)
    q, scale, zp = A._rtn_quantize(w, quant_type, strategy, group_size, symmetric, False, 1.0, False,
                                   np.dtype(np.float32), quant_type.np_dtype)
    assert q.shape == w.shape and q.dtype == quant_type.np_dtype
    qmin, qmax = quant_type.qrange(symmetric)
    assert q.min() >= qmin and q.max() <= qmax
    if strategy == QuantizationStrategy.TENSOR:
        assert scale.shape == () and zp.shape == ()
    elif strategy == QuantizationStrategy.CHANNEL:
        assert scale.shape == (48,) and zp.shape == (48,)
    else:
        assert scale.shape == (48 * 64 // group_size, 1) == zp.shape
    assert scale.dtype == np.float32 and zp.dtype == quant_type.np_dtype
    dq = F._dequantize_array(q, scale, zp, preprocess=True, strategy=strategy,
                             group_size=-1 if group_size is None else group_size)
    smax = float(np.max(scale))
    assert np.max(np.abs(dq - w)) <= 2 * smax


def test_rtn_all_zero_weights(A):
    # test_rtn.py:455-479
    for strategy, g in ((QuantizationStrategy.TENSOR, None), (QuantizationStrategy.CHANNEL, -1), (QuantizationStrategy.GROUP, 8)):
        q, s, z = A._rtn_quantize(np.zeros((16, 8), np.float32), QuantType.QInt8, strategy, g, False, False, 1.0, False,
                                  np.dtype(np.float32), np.dtype(np.int8))
        assert np.all(s == 1.0) and np.all(q == z.reshape(-1)[0])


def test_small_kernels_vs_golden(F):
    K = load_npz("kernels.npz")
    x = K["x"]
    lo = np.minimum(x.min(axis=1, keepdims=True), 0)
    hi = np.maximum(x.max(axis=1, keepdims=True), 0)
    for qtype in ("int4", "uint4", "int8", "uint8"):
        for sym in (False, True):
            for red in (False, True):
                tag = f"{qtype}_{int(sym)}{int(red)}"
                s, z = F._compute_qparams(lo, hi, QT[qtype], sym, red, np.dtype(np.float32), QT[qtype].np_dtype)
                assert s.tobytes() == K[f"qp_{tag}_s"].tobytes(), tag
                np.testing.assert_array_equal(z, K[f"qp_{tag}_z"])
                q = F._quantize_array_from_qparams(x, s, z, QT[qtype], sym, red)
                np.testing.assert_array_equal(q, K[f"qp_{tag}_q"])
                assert q.dtype == K[f"qp_{tag}_q"].dtype
                assert F._dequantize_array(q, s, z).tobytes() == K[f"qp_{tag}_dq"].tobytes()
                assert F._fake_quantize_array(x, s, z, QT[qtype], sym, red).tobytes() == K[f"qp_{tag}_dq"].tobytes()
    for sname, g in (("tensor", -1), ("channel", -1), ("group", 16)):
        out = F._dequantize_array(K[f"dq_{sname}_q"], K[f"dq_{sname}_s"], K[f"dq_{sname}_z"], preprocess=True,
                                  strategy=QuantizationStrategy(sname), group_size=g)
        assert np.ascontiguousarray(out).tobytes() == K[f"dq_{sname}_out"].tobytes()


def test_wire_format_packers():
    import torch
    from onnx_quantize_amd.hip import ops
    for vals, qtype, expected in KATS["pack"]:      # test_pack.py:11-27, :59-75
        a = torch.tensor(vals, dtype=torch.int8 if qtype == "int4" else torch.uint8, device="cuda")
        assert ops.pack_nibbles(a).cpu().tolist() == expected
    r = np.random.default_rng(0)                    # test_common.py:7-31
    z = r.integers(0, 16, size=(4 * 5, 1), dtype=np.uint8)
    pz = ops.pack_zero_points_u4(torch.from_numpy(z).cuda(), 4, 5).cpu().numpy()
    _, _, epz = O.matmul_nbits_layout(np.zeros((80, 4), np.uint8), np.zeros(20, np.float32), z, 16, 4)
    np.testing.assert_array_equal(pz, epz)


# ------------------------------------------------------------------ test_minmax_calibrator.py
class TestMinMaxCalibrator:
    def make(self, **kw):
        from onnx_quantize_amd.calibration import MinMaxCalibrator
        return MinMaxCalibrator(**kw)

    def test_collect_single_batch(self):
        c = self.make()
        c.collect("test", np.array([1.0, 2.0, 3.0, 4.0, 5.0]))
        assert "test" in c.data and c.data["test"].min_val == 1.0 and c.data["test"].max_val == 5.0
        lo, hi = c.compute_range("test")
        np.testing.assert_almost_equal(lo, 0.0)
        np.testing.assert_almost_equal(hi, 5.0)
        assert lo.dtype == np.float32 and lo.shape == ()

    def test_negative_and_multidimensional(self):
        c = self.make()
        c.collect("a", np.array([-5.0, -2.0, 0.0, 3.0, 7.0]))
        assert (c.data["a"].min_val, c.data["a"].max_val) == (-5.0, 7.0)
        c.collect("b", np.array([-10.0, -5.0, -2.0, -1.0]))
        lo, hi = c.compute_range("b")
        assert (float(lo), float(hi)) == (-10.0, 0.0)
        c.collect("c", np.array([[1.0, 2.0, 3.0], [4.0, 5.0, 6.0], [0.5, 7.0, 2.5]]))
        assert (c.data["c"].min_val, c.data["c"].max_val) == (0.5, 7.0)
        c.collect("d", np.array([42.0]))
        assert (c.data["d"].min_val, c.data["d"].max_val) == (42.0, 42.0)
        c.collect("z", np.zeros(10))
        assert (c.data["z"].min_val, c.data["z"].max_val) == (0.0, 0.0)
        assert len(c.data) == 5

    def test_multiple_batches_no_momentum(self):
        c = self.make(momentum=0.0)
        for b in ([1.0, 2.0, 3.0], [-0.5, 4.0, 2.5], [1.5, 3.5, 5.5]):
            c.collect("test", np.array(b))
        assert (c.data["test"].min_val, c.data["test"].max_val) == (-0.5, 5.5)

    def test_multiple_batches_with_momentum(self):
        c = self.make(momentum=0.8)
        c.collect("test", np.array([-1.0, 2.0, 3.0]))
        assert (c.data["test"].min_val, c.data["test"].max_val) == (-1.0, 3.0)
        c.collect("test", np.array([-0.5, 2.5, 4.0]))
        assert np.isclose(c.data["test"].min_val, -0.9) and np.isclose(c.data["test"].max_val, 3.2)
        lo, hi = c.compute_range("test")
        np.testing.assert_almost_equal(lo, -0.9)
        np.testing.assert_almost_equal(hi, 3.2)

    def test_golden_sequences(self):
        """Bit-exact running state after every batch (fp32 EMA rounding included)."""
        M = load_npz("minmax.npz")
        for seq in load_json("minmax.json"):
            sid = seq["id"]
            c = self.make(momentum=seq["momentum"])
            for b in range(seq["batches"]):
                c.collect("t", M[f"{sid}_b{b}"])
                assert np.float32(c.data["t"].min_val).tobytes() == M[f"{sid}_b{b}_min"].tobytes(), (sid, b)
                assert np.float32(c.data["t"].max_val).tobytes() == M[f"{sid}_b{b}_max"].tobytes(), (sid, b)
            lo, hi = c.compute_range("t")
            assert lo.tobytes() == M[f"{sid}_lo"].tobytes() and hi.tobytes() == M[f"{sid}_hi"].tobytes()

    def test_device_tensors_and_large_batches(self):
        """Activations already in HBM (the on-device driver of SURVEY.md 8f-N1), gemma3-shaped batch,
        odd element counts and unaligned views."""
        import torch
        c = self.make()
        gen = torch.Generator(device="cuda").manual_seed(3)
        x = torch.randn((10, 512, 640), generator=gen, device="cuda")
        c.collect("act", x)
        assert c.data["act"].min_val == x.min().item() and c.data["act"].max_val == x.max().item()
        y = torch.randn(1_000_003, generator=gen, device="cuda")
        c.collect("odd", y[1:])
        assert c.data["odd"].min_val == y[1:].min().item() and c.data["odd"].max_val == y[1:].max().item()
        c.collect("odd", y[:7])
        assert c.data["odd"].min_val == min(y[1:].min().item(), y[:7].min().item())

    @pytest.mark.parametrize("momentum", [0.0, 0.8])
    def test_collect_many_equals_per_tensor_collect(self, momentum):
        """One launch pair per calibration batch (oq_minmax_collect_many_f32) leaves exactly the state that per-tensor
        collect calls leave -- first sight, running min / max and the fp32 EMA -- for ragged sizes, unaligned views
        and names that appear only in some batches."""
        import torch
        gen = torch.Generator(device="cuda").manual_seed(5)
        a, b = self.make(momentum=momentum), self.make(momentum=momentum)
        base = torch.randn(3_000_011, generator=gen, device="cuda")
        for batch in range(4):
            tensors = {"act0": torch.randn((10, 512, 640), generator=gen, device="cuda") * (batch + 1),
                       "act1": torch.randn((7, 333), generator=gen, device="cuda") - batch,
                       "odd": base[1 + batch: 1 + batch + 1_000_001],
                       "tiny": torch.randn(3, generator=gen, device="cuda")}
            if batch % 2:
                tensors["sometimes"] = torch.randn((64, 1024), generator=gen, device="cuda")
            a.collect_many(tensors)
            for name, t in tensors.items():
                b.collect(name, t)
            for name in tensors:
                assert np.float32(a.data[name].min_val).tobytes() == np.float32(b.data[name].min_val).tobytes(), (batch, name)
                assert np.float32(a.data[name].max_val).tobytes() == np.float32(b.data[name].max_val).tobytes(), (batch, name)
        assert set(a.data) == set(b.data)
        # host arrays and fp64 tensors fall back to the per-tensor path inside collect_many
        a.collect_many({"host": np.array([1.0, -2.0, 3.0]), "f64": torch.tensor([4.0, -5.0], dtype=torch.float64, device="cuda")})
        assert (a.data["host"].min_val, a.data["host"].max_val) == (-2.0, 3.0) and a.data["f64"].min_val == -5.0

    def test_compute_qparams_many_equals_per_name_path(self):
        """One kernel + one copy for all names == compute_range + _compute_qparams per name (calibrate.py:268-285)."""
        import torch
        from onnx_quantize_amd import QuantType
        from onnx_quantize_amd.algorithms.functional import _compute_qparams
        c = self.make()
        gen = torch.Generator(device="cuda").manual_seed(8)
        data = {f"t{i}": torch.randn(1000 + 37 * i, generator=gen, device="cuda") * (i + 0.5) + (i - 3) for i in range(9)}
        data["positive"] = torch.rand(500, generator=gen, device="cuda") + 2.0      # zero must still be inside the range
        c.collect_many(data)
        for qt, sym in ((QuantType.QInt8, False), (QuantType.QUInt8, False), (QuantType.QInt8, True)):
            many = c.compute_qparams_many(list(data), qt, sym)
            for name in data:
                es, ez = _compute_qparams(*c.compute_range(name), qt, sym, False, np.float32, qt.np_dtype)
                assert many[name][0].tobytes() == es.tobytes() and int(many[name][1]) == int(ez) and many[name][1].dtype == ez.dtype
        with pytest.raises(KeyError):
            c.compute_qparams_many(["nope"], QuantType.QInt8)


def test_absmax_reductions():
    """S1: smooth_quant.py:62-74 column / row absmax."""
    import torch
    from onnx_quantize_amd.hip import ops
    gen = torch.Generator(device="cuda").manual_seed(11)
    x = torch.randn((6, 333, 644), generator=gen, device="cuda") * 3
    got = ops.absmax(x)
    assert torch.equal(got, x.reshape(-1, 644).abs().amax(dim=0))
    w = torch.randn((515, 1030), generator=gen, device="cuda")
    assert torch.equal(ops.absmax(w, per_row=True), w.abs().amax(dim=1))
    assert torch.equal(ops.absmax(w[:, 1:1022]), w[:, 1:1022].abs().amax(dim=0))
    np.testing.assert_array_equal(got.cpu().numpy(), O.absmax_cols(x.cpu().numpy()))


# ------------------------------------------------------------------ test_gptq.py
@pytest.mark.parametrize("group_size", [8, 16, 64, -1])
@pytest.mark.parametrize("block_size", [32, 128, 256])
@pytest.mark.parametrize("actorder", [True, False])
def test_gptq_quantize_grid(A, rng, block_size, group_size, actorder):
    w = rng.normal(0, 1, (16, 32)).astype(np.float32)
    x = rng.normal(0, 1, (32, 16)).astype(np.float32)
    w_q, w_scale, w_zp = A._gptq_quantize(w, x, group_size=group_size, strategy=QuantizationStrategy.TENSOR,
                                          block_size=block_size, percdamp=0.01, actorder=actorder)
    assert w_q.shape == w.shape and w_q.dtype == np.int8
    assert isinstance(w_scale, np.ndarray) and w_scale.dtype == np.float32 and w_zp.dtype == np.int8
    eq, es, ez = O.gptq_quantize(w, x, "int8", "tensor", group_size, block_size=block_size, actorder=actorder)
    np.testing.assert_array_equal(w_q, eq)
    np.testing.assert_array_equal(w_zp, ez)
    np.testing.assert_allclose(w_scale, es, rtol=1e-5)


@pytest.mark.parametrize("reduce_range", [True, False])
@pytest.mark.parametrize("clip_ratio", [0.9, 1.0])
def test_gptq_reduce_range_clip_ratio(A, rng, reduce_range, clip_ratio):
    w = rng.normal(0, 1, (16, 32)).astype(np.float32)
    x = rng.normal(0, 1, (32, 16)).astype(np.float32)
    w_q, _, _ = A._gptq_quantize(w, x, strategy=QuantizationStrategy.TENSOR, reduce_range=reduce_range,
                                 clip_ratio=clip_ratio)
    lim = (-64, 64) if reduce_range else (-128, 127)
    assert w_q.min() >= lim[0] and w_q.max() <= lim[1]


def test_gptq_streamed_batches_and_fallback_warning(A, rng, caplog):
    w = rng.normal(0, 1, (64, 40)).astype(np.float32)
    x = rng.normal(0, 1, (12, 9, 64)).astype(np.float32)
    q1, s1, z1 = A._gptq_quantize(w, x, QuantType.QInt4, QuantizationStrategy.GROUP, 32)
    q2, s2, z2 = A._gptq_quantize(w, [x[:5], x[5:]], QuantType.QInt4, QuantizationStrategy.GROUP, 32)
    np.testing.assert_array_equal(q1, q2)
    assert s1.shape == (40 * 2, 1) and s1.tobytes() == s2.tobytes()
    import logging
    with caplog.at_level(logging.WARNING):
        logging.getLogger("onnx_quantize_amd.algorithms.gptq").propagate = True
        A._gptq_quantize(w, x, percdamp=-2.0)          # negative damping -> not positive definite
    assert "Falling back to round-to-nearest" in caplog.text


def test_calibration_of_a_tensor_with_more_than_2_31_elements():
    """minmax.py:40-64 on 2.2e9 activations (8.8 GB): extremes planted at the very first and very last element, one
    tensor through `collect` and through `collect_many` together with a small one."""
    import torch
    from onnx_quantize_amd.calibration import MinMaxCalibrator
    n = 2**31 + 2**26 + 12345
    x = torch.empty(n, device="cuda")
    for i in range(0, n, 2**28):
        x[i:i + 2**28].uniform_(-1.0, 1.0)
    x[0], x[-1], x[2**31 + 7] = -7.25, 9.5, 3.0
    cal = MinMaxCalibrator()
    cal.collect("big", x)
    assert cal.data["big"].min_val == np.float32(-7.25) and cal.data["big"].max_val == np.float32(9.5)
    small = torch.linspace(-2, 11, 1000, device="cuda")
    many = MinMaxCalibrator()
    many.collect_many({"big": x, "small": small})
    assert many.data["big"].min_val == np.float32(-7.25) and many.data["big"].max_val == np.float32(9.5)
    assert many.data["small"].min_val == np.float32(-2) and many.data["small"].max_val == np.float32(11)
    del x
    torch.cuda.empty_cache()


def test_calibrator_propagates_nan_like_numpy():
    """minmax.py:47-48 uses np.min / np.max: one NaN activation makes the tensor's range NaN (and keeps it NaN through the
    running min / max and the EMA).  ADVICE r01: the reductions used to drop NaN and hide the corruption."""
    from onnx_quantize_amd.calibration import MinMaxCalibrator
    rng = np.random.default_rng(12)
    clean = rng.standard_normal((4, 33, 65)).astype(np.float32)
    dirty = clean.copy()
    dirty[2, 17, 40] = np.nan
    for momentum in (0.0, 0.5):
        cal, ref = MinMaxCalibrator(momentum), O.MinMaxOracle(momentum)
        for x in (clean, dirty, clean):
            cal.collect("x", x)
            cal.collect_many({"y": x})
            ref.collect("x", x)
        assert np.isnan(ref.data["x"][0]) and np.isnan(ref.data["x"][1])          # what NumPy (the reference) ends up with
        for name in ("x", "y"):
            assert np.isnan(cal.data[name].min_val) and np.isnan(cal.data[name].max_val), (momentum, name)
        ok = MinMaxCalibrator(momentum)
        ok.collect("x", clean)
        assert np.isfinite(ok.data["x"].min_val)


def test_column_shard_kernels_equal_the_whole_matrix_path():
    """sharding.HipKernels on one GPU: the per-tensor path built from the calibration min / max
    kernel + oq_qparams_f32 + oq_quantize_f32 gives the bits of the fused two-pass kernel, also when the matrix is cut into
    column shards whose ranges are combined the way the all_reduce combines them; group / channel shards equal the
    corresponding columns of the whole result; column-sharded GPTQ equals GPTQ of the whole matrix."""
    import torch
    from onnx_quantize_amd import sharding as S
    from onnx_quantize_amd.hip import ops
    gen = torch.Generator(device="cuda").manual_seed(3)
    w = torch.randn((512, 320), generator=gen, device="cuda")
    w[:, 300:] *= 4
    ranges = S.column_ranges(320, 3, 32)
    assert ranges == [(0, 128), (128, 224), (224, 320)]
    for qtype, sym, clip in (("uint8", False, 1.0), ("int8", True, 0.9), ("uint4", False, 0.95)):
        q, s, z = ops.rtn_quantize(w, qtype, "tensor", -1, sym, False, clip)
        lq, ls, lz = S.rtn_quantize_column_shard(w, qtype, "tensor", -1, sym, False, clip)
        assert torch.equal(lq, q) and torch.equal(ls, s) and torch.equal(lz, z) and ls.shape == s.shape == ()
        # what the exchange does: (-min, max) of every shard under MAX
        mm = torch.stack([torch.stack([-m[0], m[1]]) for m in (S.HipKernels.minmax(w[:, a:b].contiguous()) for a, b in ranges)]).amax(0)
        zero = torch.zeros((), device="cuda")
        lo, hi = torch.minimum(-mm[0] * clip, zero), torch.maximum(mm[1] * clip, zero)
        for a, b in ranges:
            sq, ss, sz = S.HipKernels.quantize_tensor(w[:, a:b].contiguous(), lo, hi, qtype, sym, False)
            assert torch.equal(sq, q[:, a:b]) and torch.equal(ss, s) and torch.equal(sz, z)
    q, s, z = ops.rtn_quantize(w, "uint4", "group", 128)
    for a, b in ranges:
        sq, ss, sz = S.rtn_quantize_column_shard(w[:, a:b].contiguous(), "uint4", "group", 128)
        assert torch.equal(sq, q[:, a:b]) and torch.equal(ss, s[a * 4:b * 4]) and torch.equal(sz, z[a * 4:b * 4])
    x = [torch.randn((2, 64, 512), generator=gen, device="cuda") for _ in range(2)]
    h = torch.zeros((512, 512), device="cuda")
    n = 0
    for xb in x:
        n = ops.hessian_accumulate(xb, h, n)
    gq, gs, gz, _ = ops.gptq_quantize(w, h, "int4", "group", 128)
    for a, b in ranges:
        sq, ss, sz = S.gptq_quantize_column_shard(w[:, a:b].contiguous(), x, "int4", "group", 128)
        assert torch.equal(sq, gq[:, a:b]) and torch.equal(sz, gz[a * 4:b * 4]) and torch.equal(ss, gs[a * 4:b * 4])


def test_plugin_seam_against_the_reference_plugins():
    """tests/golden/seam.*: `qconfig.weights.algorithm.quantize_weights(w, qconfig, out=out)` -- the one call through which the
    reference reaches the path (qrules/_common.py:133) -- of the reference's RTNConfig / GPTQConfig / HqqConfig on carrier
    values, against the same call on this package's configs: shapes, dtypes, integers and zero points identical, scales
    bit-equal (RTN, GPTQ parity mode) or within the HQQ tolerance."""
    import types
    from onnx_quantize_amd import GPTQConfig, HqqConfig, QConfig, QWeightArgs
    G, cases = load_npz("seam.npz"), load_json("seam.json")["cases"]
    assert len(cases) == 8
    algos = {"rtn": None, "gptq": GPTQConfig, "hqq": HqqConfig}
    for c in cases:
        key = c["key"]
        kw = {**c["weights"], "dtype": QuantType.from_string(c["weights"]["dtype"])}
        if algos[c["algorithm"]] is not None:
            kw["algorithm"] = algos[c["algorithm"]](**c["config"])
        qc = QConfig(weights=QWeightArgs(**kw))
        w = types.SimpleNamespace(name="w", const_value=types.SimpleNamespace(numpy=lambda a=G[key + "_w"]: a))
        node = types.SimpleNamespace(meta={"input": G[key + "_x"].copy()})
        out = types.SimpleNamespace(producer=lambda node=node: node)
        q, s, z = qc.weights.algorithm.quantize_weights(w, qc, out=out)
        assert list(np.shape(q)) == c["q_shape"] and list(np.shape(s)) == c["s_shape"] and list(np.shape(z)) == c["z_shape"], key
        assert str(np.asarray(s).dtype) == c["s_dtype"] and str(np.asarray(z).dtype) == c["z_dtype"], key
        eq, es, ez = G[key + "_q"], G[key + "_s"], G[key + "_z"]
        if c["algorithm"] == "hqq":
            assert np.asarray(s).tobytes() == es.tobytes()
            np.testing.assert_allclose(np.asarray(z, np.float32), ez, atol=2e-5)
            d = np.abs(np.asarray(q).astype(np.int32) - eq.astype(np.int32))
            assert d.max() <= 1 and np.count_nonzero(d) <= 2e-3 * d.size
        else:
            np.testing.assert_array_equal(np.asarray(q).astype(eq.dtype), eq)
            np.testing.assert_array_equal(np.asarray(z).astype(np.int32), ez)
            np.testing.assert_allclose(np.asarray(s), es, rtol=1e-5, atol=0)
        np.testing.assert_array_equal(node.meta["input"], G[key + "_x"])          # inputs are never mutated


def test_random_calibration_sequences_against_the_oracle():
    """Property test (hypothesis, fixed seed): random sequences of activation tensors (ranks, sizes incl. unaligned odd ones,
    float32 / float64, several names, momentum 0 or not) through `collect` and `collect_many`: the running state and the
    range handed out equal the oracle's (= minmax.py:40-87) bit for bit, EMA included."""
    import torch
    from hypothesis import HealthCheck, given, seed, settings, strategies as st
    from onnx_quantize_amd.calibration import MinMaxCalibrator

    @st.composite
    def case(draw):
        momentum = draw(st.sampled_from([0.0, 0.0, 0.9, 0.5, 0.01]))
        steps = draw(st.lists(st.tuples(st.sampled_from(["a", "b", "c"]), st.lists(st.integers(1, 40), min_size=1, max_size=3),
                                        st.sampled_from(["f32", "f32", "f64"]), st.integers(0, 3)), min_size=1, max_size=8))
        return momentum, steps, draw(st.booleans()), draw(st.integers(0, 2**31 - 1))

    @seed(20240603)
    @settings(max_examples=60, deadline=None, suppress_health_check=list(HealthCheck))
    @given(case())
    def run(c):
        momentum, steps, many, rs = c
        r = np.random.default_rng(rs)
        cal, ref = MinMaxCalibrator(momentum), O.MinMaxOracle(momentum)
        dtypes = {}
        for name, shape, dt, off in steps:
            dt = dtypes.setdefault(name, dt)                      # one dtype per name, like a real graph value
            x = (r.standard_normal(shape) * r.uniform(0.1, 50)).astype(np.float32 if dt == "f32" else np.float64)
            ref.collect(name, x)
            t = torch.from_numpy(np.concatenate([np.zeros(off, x.dtype), x.ravel()])).cuda()[off:].reshape(shape)   # unaligned base
            if many and dt == "f32":
                cal.collect_many({name: t})
            else:
                cal.collect(name, t)
        for name in dtypes:
            lo, hi = ref.data[name]
            assert cal.data[name].min_val == lo and cal.data[name].max_val == hi, (c, name)
            a, b = cal.compute_range(name), ref.compute_range(name)
            assert a[0].tobytes() == b[0].tobytes() and a[1].tobytes() == b[1].tobytes()

    run()
