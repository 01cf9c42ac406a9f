"""AWQ / SmoothQuant numeric cores (reference pre_passes/awq.py, smooth_quant.py).

The oracle's restatements are pinned by tests/golden/awq.* -- the reference's own `_apply_awq`, `_apply_awq_clip` and
`_smooth_quant_node` run on six small layers (make_golden.py::gen_awq) -- and the GPU composition is checked against
that oracle.
Tolerances: reductions run in a different order on the device, so the search scale differs by a few ulp, an RTN
integer may flip, and the 20 / 10 losses agree to 2e-3 relative; the chosen grid point must be the oracle's or one
whose oracle loss is within that tolerance of the oracle's minimum."""
import numpy as np
import pytest

import oq_oracle as O
from onnx_quantize_amd import AwqConfig, QConfig, QWeightArgs, QuantType, SmoothQuantConfig


def test_configs_mirror_the_reference():
    a, s = AwqConfig(), SmoothQuantConfig()
    assert (a.preprocessing_type, a.clip_search) == ("awq", False)
    assert (s.preprocessing_type, s.alpha) == ("smooth_quant", 0.5)
    q = QConfig(weights=QWeightArgs(dtype="uint4", strategy="group", group_size=32), preprocessors=[{"preprocessing_type": "awq", "clip_search": True}],
                calibration_data=np.zeros((2, 4), np.float32))
    assert isinstance(q.preprocessors[0], AwqConfig) and q.preprocessors[0].clip_search
    with pytest.raises(ImportError, match="delegated to the reference package"):
        a.build_pass(q)


def test_oracle_reproduces_what_the_reference_passes_computed():
    """tests/golden/awq.*: `_apply_awq`, `_apply_awq_clip` (pre_passes/awq.py:114-259) and `_smooth_quant_node`
    (smooth_quant.py:91-129) run unmodified on small layers (make_golden.py::gen_awq); the oracle's restatements give the
    same chosen scale (as the emitted Mul constant 1 / scale), the same rescaled weights and activations, the same clip
    ratio -- bit for bit."""
    from conftest import load_json, load_npz
    G, cases = load_npz("awq.npz"), load_json("awq.json")["cases"]
    assert len(cases) == 6
    for c in cases:
        key, g = c["key"], c["group_size"]
        x, w = G[key + "_x"], G[key + "_w"]
        best, losses = O.awq_scale_search(x, w, c["qtype"], c["strategy"], g, c["symmetric"])
        assert best.dtype == np.float32 and len(losses) == 20
        np.testing.assert_array_equal(1.0 / best, G[key + "_awq_inv_scale"])
        np.testing.assert_array_equal(w * best.reshape(-1, 1), G[key + "_awq_w"])
        np.testing.assert_array_equal(x / best.reshape(1, -1), G[key + "_awq_x"])
        ratio, _ = O.awq_clip_search(x, w, c["qtype"], c["strategy"], g, c["symmetric"])
        assert ratio == c["clip_ratio"]
        for alpha in (0.5, 0.8):
            sc = O.smooth_quant_scale(x, w, alpha)
            np.testing.assert_array_equal(1.0 / sc, G[key + f"_sq{int(alpha * 10)}_inv_scale"])
            np.testing.assert_array_equal(np.multiply(sc.reshape(-1, 1), w), G[key + f"_sq{int(alpha * 10)}_w"])


def test_oracle_search_properties(rng):
    """awq.py:143-184: ratio 0 leaves only the weight term; the winning scale is normalised (sqrt(max * min) == 1);
    the clip search returns a ratio of its grid; SmoothQuant with alpha = 1 is the activation absmax."""
    x = rng.standard_normal((3, 8, 32)).astype(np.float32) * rng.uniform(0.1, 4, 32).astype(np.float32)
    w = rng.standard_normal((32, 24)).astype(np.float32)
    scale, losses = O.awq_scale_search(x, w, "uint4", "group", 16)
    assert losses.shape == (20,) and np.isclose(np.sqrt(scale.max() * scale.min()), 1.0, rtol=1e-6)
    ratio, l2 = O.awq_clip_search(x, w, "uint4", "group", 16)
    assert ratio in [1 - i / 100 for i in range(10)] and l2.shape == (10,)
    np.testing.assert_allclose(O.smooth_quant_scale(x, w, 1.0), np.maximum(np.abs(x.reshape(-1, 32)).max(0), 1e-5), rtol=1e-6)


def _inputs(seed, t, k, n):
    r = np.random.default_rng(seed)
    x = r.standard_normal((4, t // 4, k)).astype(np.float32) * r.uniform(0.1, 4, k).astype(np.float32)
    w = (r.standard_normal((k, n)) * 0.05).astype(np.float32)
    return x, w


@pytest.mark.gpu
@pytest.mark.parametrize("qtype,strategy,g,sym", [("uint4", "group", 32, False), ("int8", "channel", -1, True), ("uint8", "tensor", -1, False),
                                                  ("int4", "group", 128, False), ("int8", "group", 64, True), ("uint4", "group", 16, True),
                                                  ("int4", "group", 256, True),
                                                  # groups below the fused kernel's 16 rows (the reference's own AWQ tests use 8,
                                                  # test/pre_passes/test_awq.py:68): general route, vector and scalar residual kernels
                                                  ("int8", "group", 8, False), ("uint4", "group", 4, False), ("int4", "group", 8, True)])
def test_gpu_awq_searches_follow_the_oracle(qtype, strategy, g, sym):
    from onnx_quantize_amd.preprocessing import awq_clip_search, awq_scale_search
    x, w = _inputs(3, 512, 256, 192)
    es, el = O.awq_scale_search(x, w, qtype, strategy, g, sym)
    s, l = awq_scale_search(x, w, QuantType.from_string(qtype), strategy, g, sym)
    np.testing.assert_allclose(l, el, rtol=2e-3)
    assert el[int(np.argmin(l))] <= el.min() * (1 + 2e-3)
    if int(np.argmin(l)) == int(np.argmin(el)):
        np.testing.assert_allclose(s, es, rtol=1e-5)
    er, ecl = O.awq_clip_search(x, w, qtype, strategy, g, sym)
    r, cl = awq_clip_search(x, w, QuantType.from_string(qtype), strategy, g, sym)
    np.testing.assert_allclose(cl, ecl, rtol=2e-3)
    assert ecl[int(round((1 - r) * 100))] <= ecl.min() * (1 + 2e-3)


@pytest.mark.gpu
@pytest.mark.parametrize("t,k,n,qtype,g", [(300, 512, 1000, "uint4", 64), (96, 256, 260, "int8", 128), (640, 1024, 768, "uint4", 16), (200, 176, 512, "uint4", 16)])
def test_gpu_awq_searches_on_the_fused_pieces_route_follow_the_oracle(t, k, n, qtype, g):
    """Round 5: with groups of 16-128 rows the quantize-residual kernel writes the loss product's first fp16 pieces itself
    (awq.hip: no fp32 D, no split launch), scaled by an upper BOUND of |D| instead of its maximum.  Widths that are not a
    multiple of the 256-column tile (zero-filled padding of the pieces), outlier channels (a loose bound: candidate scales that
    span two orders of magnitude inside a group) and K = 176 (not a whole number of stages: the unfused route) -- same bars."""
    from onnx_quantize_amd.preprocessing import awq_clip_search, awq_scale_search
    x, w = _inputs(5 + n, t, k, n)
    x[..., ::29] *= 80.0
    es, el = O.awq_scale_search(x, w, qtype, "group", g)
    s, l = awq_scale_search(x, w, QuantType.from_string(qtype), "group", g)
    np.testing.assert_allclose(l, el, rtol=2e-3)
    assert el[int(np.argmin(l))] <= el.min() * (1 + 2e-3)
    if int(np.argmin(l)) == int(np.argmin(el)):
        np.testing.assert_allclose(s, es, rtol=1e-5)
    er, ecl = O.awq_clip_search(x, w, qtype, "group", g)
    r, cl = awq_clip_search(x, w, QuantType.from_string(qtype), "group", g)
    np.testing.assert_allclose(cl, ecl, rtol=2e-3)
    assert ecl[int(round((1 - r) * 100))] <= ecl.min() * (1 + 2e-3)


@pytest.mark.gpu
@pytest.mark.parametrize("t,k,n,qtype,strategy,g", [(2048, 256, 192, "uint4", "group", 32), (8192, 1024, 516, "uint4", "group", 128),
                                                    (6144, 1024, 256, "int8", "channel", -1), (1024, 128, 64, "uint8", "tensor", -1)])
def test_gpu_awq_searches_with_long_calibration_sets_follow_the_oracle(t, k, n, qtype, strategy, g):
    """T >= 3 K rows: the searches take the Gram route (awq.hip: sum_n ||X d_n||^2 = <D, X^T X D>, the Gram matrix made once
    by the Hessian kernels -- fp32 MFMA below K = 1024, fp16 pieces above).  Same bars as the direct route; outlier channels
    make the Gram matrix span six orders of magnitude."""
    from onnx_quantize_amd.preprocessing import awq_clip_search, awq_scale_search
    x, w = _inputs(11 + k, t, k, n)
    x[..., ::37] *= 40.0
    es, el = O.awq_scale_search(x, w, qtype, strategy, g)
    s, l = awq_scale_search(x, w, QuantType.from_string(qtype), strategy, g)
    np.testing.assert_allclose(l, el, rtol=2e-3)
    assert el[int(np.argmin(l))] <= el.min() * (1 + 2e-3)
    if int(np.argmin(l)) == int(np.argmin(el)):
        np.testing.assert_allclose(s, es, rtol=1e-5)
    er, ecl = O.awq_clip_search(x, w, qtype, strategy, g)
    r, cl = awq_clip_search(x, w, QuantType.from_string(qtype), strategy, g)
    np.testing.assert_allclose(cl, ecl, rtol=2e-3)
    assert ecl[int(round((1 - r) * 100))] <= ecl.min() * (1 + 2e-3)


@pytest.mark.gpu
def test_gpu_smooth_quant_scale_matches_oracle():
    from onnx_quantize_amd.preprocessing import smooth_quant_scale
    x, w = _inputs(4, 1024, 384, 128)
    x[..., 7] = 0.0                                   # a dead activation channel: clamped to 1e-5 (smooth_quant.py:66-67)
    for alpha in (0.5, 0.25, 1.0):
        np.testing.assert_allclose(smooth_quant_scale(x, w, alpha), O.smooth_quant_scale(x, w, alpha), rtol=2e-6)


@pytest.mark.gpu
def test_gpu_searches_against_the_reference_passes_outputs():
    """The GPU searches on the six golden layers (tests/golden/awq.*): the SmoothQuant scale to 1e-6; the AWQ scale
    equals the one the reference's `_apply_awq` emitted whenever the same grid point wins, and the winning grid point is
    the reference's or one whose (oracle) loss is within 2e-3 of it; the clip ratio likewise."""
    from conftest import load_json, load_npz
    from onnx_quantize_amd.preprocessing import awq_clip_search, awq_scale_search, smooth_quant_scale
    G, cases = load_npz("awq.npz"), load_json("awq.json")["cases"]
    for c in cases:
        key, g = c["key"], c["group_size"]
        x, w = G[key + "_x"], G[key + "_w"]
        qt = QuantType.from_string(c["qtype"])
        s, losses = awq_scale_search(x, w, qt, c["strategy"], g, c["symmetric"])
        _, el = O.awq_scale_search(x, w, c["qtype"], c["strategy"], g, c["symmetric"])
        assert el[int(np.argmin(losses))] <= el.min() * (1 + 2e-3)
        if int(np.argmin(losses)) == int(np.argmin(el)):
            np.testing.assert_allclose(1.0 / s, G[key + "_awq_inv_scale"], rtol=2e-5)
        r, cl = awq_clip_search(x, w, qt, c["strategy"], g, c["symmetric"])
        _, ecl = O.awq_clip_search(x, w, c["qtype"], c["strategy"], g, c["symmetric"])
        assert ecl[int(round((1 - r) * 100))] <= ecl.min() * (1 + 2e-3)
        if int(np.argmin(cl)) == int(np.argmin(ecl)):
            assert r == c["clip_ratio"]
        for alpha in (0.5, 0.8):
            np.testing.assert_allclose(1.0 / smooth_quant_scale(x, w, alpha), G[key + f"_sq{int(alpha * 10)}_inv_scale"], rtol=2e-6)


@pytest.mark.gpu
@pytest.mark.parametrize("qtype,strategy,g,sym", [("uint4", "group", 32, False), ("int8", "channel", -1, True), ("uint8", "tensor", -1, False),
                                                  ("int4", "group", 128, False), ("int8", "group", 8, False)])
def test_gpu_searches_from_streamed_statistics_follow_the_oracle(qtype, strategy, g, sym):
    """`ops.SearchStatistics` (running Gram matrix, |x| sums and maxima over batches) in place of the activations themselves:
    the losses are the oracle's on the concatenated batches, the SmoothQuant scale is the array route's, and `divide(scale)` is
    the reference's in-place rescale of the stored input (awq.py:191)."""
    import torch

    from onnx_quantize_amd.hip import ops
    x, w = _inputs(5, 768, 256, 192)
    xd, wd = torch.from_numpy(x).cuda(), torch.from_numpy(w).cuda()
    stats = ops.SearchStatistics(256, "cuda")
    for part in (xd[:1], xd[1:3], xd[3:]):                                  # three uneven batches
        stats.add(part)
    assert stats.rows == 768
    es, el = O.awq_scale_search(x, w, qtype, strategy, g, sym)
    s, l = ops.awq_scale_search_stats(stats, wd, qtype, strategy, g, sym)
    np.testing.assert_allclose(l, el, rtol=2e-3)
    assert el[int(np.argmin(l))] <= el.min() * (1 + 2e-3)
    if int(np.argmin(l)) == int(np.argmin(el)):
        np.testing.assert_allclose(s.cpu().numpy(), es, rtol=2e-5)
    er, ecl = O.awq_clip_search(x, w, qtype, strategy, g, sym)
    r, cl = ops.awq_clip_search_stats(stats, wd, qtype, strategy, g, sym)
    np.testing.assert_allclose(cl, ecl, rtol=2e-3)
    assert ecl[int(round((1 - r) * 100))] <= ecl.min() * (1 + 2e-3)
    assert torch.equal(ops.smooth_quant_scale_stats(stats, wd, 0.5), ops.smooth_quant_scale(xd, wd, 0.5))
    # the in-place rescale: statistics of x / scale
    scale = torch.rand(256, device="cuda") + 0.5
    scaled = ops.SearchStatistics(256, "cuda")
    scaled.add(xd / scale)
    stats.divide(scale)
    torch.testing.assert_close(stats.abs_sum, scaled.abs_sum, rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(stats.absmax, scaled.absmax, rtol=1e-6, atol=0)
    torch.testing.assert_close(stats.gram, scaled.gram, rtol=2e-4, atol=2e-5 * float(scaled.gram.abs().max()))
