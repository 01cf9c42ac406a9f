"""GPU parity of the GPTQ path (G1-G4) through the C ABI.

* Hessian (MFMA SYRK) and the inverse factor go through different fp32 operation orders than
  OpenBLAS/LAPACK: compared to 1e-5 of the matrix scale (north-star tolerance), not bit for bit.
* The emitted integers / zero points of the reference-as-written ("parity") mode are bit-exact against
  golden vectors produced by the reference itself; scales to 1e-5 relative.
"""
import numpy as np
import pytest

import oq_oracle as O
from conftest import load_json, load_npz

pytestmark = pytest.mark.gpu

GPTQ_CASES = load_json("gptq.json")
GPTQ = load_npz("gptq.npz")


@pytest.fixture(scope="module")
def ops():
    import torch
    from onnx_quantize_amd.hip import ops as _ops
    assert torch.cuda.is_available()
    return _ops


def dev(a):
    import torch
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def hessian_of(ops, x):
    import torch
    k = x.shape[-1]
    h = torch.zeros((k, k), dtype=torch.float32, device="cuda")
    n = ops.hessian_accumulate(dev(x), h, 0)
    return h, n


@pytest.mark.parametrize("method", ["auto", "f32", "bf16x6", "bf16x9", "f16x3"])
def test_hessian_vs_golden(ops, method):
    """H as the reference itself produced it (tests/golden/make_golden.py ran gptq.py:246-260), for every kernel."""
    before = ops.hessian_method()
    ops.hessian_set_method(method)
    try:
        _hessian_vs_golden(ops)
    finally:
        ops.hessian_set_method(before)


def _hessian_vs_golden(ops):
    x = GPTQ["b_x"]
    h, n = hessian_of(ops, x)
    assert n == int(GPTQ["b_nsamples"])
    ref = GPTQ["b_h"]
    got = h.cpu().numpy()
    np.testing.assert_allclose(got, ref, rtol=0, atol=1e-5 * float(np.abs(ref).max()))
    np.testing.assert_array_equal(got, got.T)                    # exactly symmetric by construction
    assert np.count_nonzero(np.diag(got) == 0) == 3              # dead channels stay exact zeros


def test_hessian_streaming_matches_one_shot(ops):
    """Batches streamed one by one (running-average algebra of gptq.py:254-258) == all at once."""
    import torch
    x = GPTQ["b_x"]
    k = x.shape[-1]
    h = torch.zeros((k, k), dtype=torch.float32, device="cuda")
    n = 0
    for i in range(0, x.shape[0], 2):
        n = ops.hessian_accumulate(dev(x[i:i + 2]), h, n)
    assert n == x.shape[0]
    ref = GPTQ["b_h"]
    np.testing.assert_allclose(h.cpu().numpy(), ref, rtol=0, atol=2e-5 * float(np.abs(ref).max()))


@pytest.mark.parametrize("t,k", [(1000, 384), (333, 200), (64, 130), (4096, 1024)])
def test_hessian_shapes_vs_float64(ops, t, k):
    rng = np.random.default_rng(t + k)
    x = rng.standard_normal((4, t // 4, k), dtype=np.float32) * rng.uniform(0.1, 3, size=k).astype(np.float32)
    h, n = hessian_of(ops, x)
    x2 = x.reshape(-1, k).astype(np.float64)
    ref = (2.0 / 4) * x2.T @ x2
    np.testing.assert_allclose(h.cpu().numpy(), ref, rtol=0, atol=2e-5 * float(np.abs(ref).max()))


@pytest.fixture
def restore_hessian_method(ops):
    before = ops.hessian_method()
    yield
    ops.hessian_set_method(before)


@pytest.mark.parametrize("method", ["f32", "bf16x6", "bf16x9", "f16x3"])
@pytest.mark.parametrize("t,k,ld", [(4096, 1024, 1024), (1000, 384, 384), (333, 200, 256), (2050, 1301, 1301), (16, 1, 1)])
def test_hessian_methods_vs_float64(ops, restore_hessian_method, method, t, k, ld):
    """Every X^T X kernel (include/oq_hip.h, G1 methods): fp32-grade against float64, exactly symmetric, exact zeros for
    dead channels, rows of X with a leading dimension, T and K that are no multiples of the 16-row stage / 256 tile."""
    import torch
    rng = np.random.default_rng(t + k)
    x = rng.standard_normal((t, ld), dtype=np.float32) * rng.uniform(0.1, 3, size=ld).astype(np.float32) + 0.25
    if k > 8:
        x[:, 5] = 0
    xd = dev(x)[:, :k]                                               # a strided view when ld > k
    ops.hessian_set_method(method)
    assert ops.hessian_method() == method
    h = torch.zeros((k, k), dtype=torch.float32, device="cuda")
    n = ops.hessian_accumulate(xd.reshape(2, t // 2, k) if ld == k else xd, h, 0)
    n_add = 2 if ld == k else t
    assert n == n_add
    x2 = x[:, :k].astype(np.float64)
    ref = (2.0 / n_add) * x2.T @ x2
    got = h.cpu().numpy()
    np.testing.assert_allclose(got, ref, rtol=0, atol=1e-5 * float(np.abs(ref).max()))
    np.testing.assert_array_equal(got, got.T)
    if k > 8:
        assert np.all(got[5] == 0) and np.all(got[:, 5] == 0)
    # second batch: the running average H n/(n+b) + (2/(n+b)) X^T X
    n2 = ops.hessian_accumulate(xd.reshape(2, t // 2, k) if ld == k else xd, h, n)
    assert n2 == 2 * n_add
    np.testing.assert_allclose(h.cpu().numpy(), ref, rtol=0, atol=1e-5 * float(np.abs(ref).max()))   # same X twice: same mean


@pytest.mark.parametrize("magnitude", [1e-17, 1e-12, 1.0, 3e4, 1e15])
def test_hessian_f16_pieces_are_scale_free(ops, restore_hessian_method, magnitude):
    """OQ_HESSIAN_F16X3 scales the batch by a power of two into fp16's range and undoes it exactly: the result relative to
    max |H| does not depend on the magnitude of the activations (H from 1e-30 to 1e37, i.e. wherever fp32 can hold it), channels 1e5 x
    apart keep their own precision where fp16's subnormals allow, an all-zero batch gives exact zeros, and the error
    against float64 is of the size of the bf16x6 kernel's (<= 1e-5 max |H|; measured 1-3e-6)."""
    import torch
    t, k = 2048, 1024
    rng = np.random.default_rng(17)
    chan = np.exp(rng.uniform(np.log(1e-3), np.log(1e2), size=k)).astype(np.float32)     # outlier channels 1e5 apart
    base = rng.standard_normal((t, k)).astype(np.float32) * chan
    base[:, 7] = 0
    x = (base.astype(np.float64) * magnitude).astype(np.float32)
    ops.hessian_set_method("f16x3")
    h = torch.zeros((k, k), device="cuda")
    n = ops.hessian_accumulate(dev(x).reshape(4, t // 4, k), h, 0)
    assert n == 4
    x64 = x.astype(np.float64)
    ref = (2.0 / 4) * x64.T @ x64
    got = h.cpu().numpy().astype(np.float64)
    assert np.isfinite(got).all()
    assert np.abs(got - ref).max() <= 1e-5 * np.abs(ref).max()
    np.testing.assert_array_equal(got, got.T)
    assert np.all(got[7] == 0) and np.all(got[:, 7] == 0)
    # a mid-sized channel pair (both ~1e-2 of the largest) keeps fp32-grade RELATIVE accuracy
    mid = np.argsort(chan)[k // 2: k // 2 + 8]
    sub_ref, sub = ref[np.ix_(mid, mid)], got[np.ix_(mid, mid)]
    assert np.abs(sub - sub_ref).max() <= 2e-5 * np.abs(sub_ref).max()
    z = torch.zeros((k, k), device="cuda")
    ops.hessian_accumulate(torch.zeros((2, 64, k), device="cuda"), z, 0)
    assert float(z.abs().max()) == 0.0


def test_hessian_f16_pieces_keep_dead_channels_the_reference_s(ops, restore_hessian_method):
    """gptq.py:284-286 calls a channel dead when H[k][k] == 0.  A channel 2^-45 below the batch maximum lies under fp16's
    last subnormal after the scaling: its samples keep one unit there, so H[k][k] > 0 as on the fp32 path; an all-zero
    channel stays exactly zero, and so does one whose squares vanish in fp32 too."""
    import torch
    t, k = 2048, 1024
    rng = np.random.default_rng(5)
    x = rng.standard_normal((t, k)).astype(np.float32)
    x[:, 3] *= np.float32(2.0 ** -45)          # alive in fp32 (x^2 ~ 2^-90), below 2^-24 after the fp16 scaling
    x[:, 4] = 0                                # dead
    x[:, 5] *= np.float32(1e-30)               # x^2 underflows in fp32 as well: dead for the reference too
    xd = torch.from_numpy(x).cuda().reshape(4, t // 4, k)
    diag = {}
    for m in ("f32", "f16x3"):
        ops.hessian_set_method(m)
        h = torch.zeros((k, k), device="cuda")
        ops.hessian_accumulate(xd, h, 0)
        diag[m] = torch.diagonal(h).cpu().numpy()
    for m in diag:
        assert diag[m][3] > 0 and diag[m][4] == 0 and diag[m][5] == 0, m
    np.testing.assert_array_equal(diag["f32"] == 0, diag["f16x3"] == 0)


def test_hessian_auto_picks_the_split_kernel_for_wide_inputs_only(ops, restore_hessian_method):
    import torch
    ops.hessian_set_method("auto")
    x = torch.randn((2, 1024, 1024), device="cuda")                  # K = 1024 needs T >= 2048 rows for the split kernel
    outs = {}
    for m in ("auto", "f16x3", "f32"):
        ops.hessian_set_method(m)
        h = torch.zeros((1024, 1024), device="cuda")
        ops.hessian_accumulate(x, h, 0)
        outs[m] = h
    assert torch.equal(outs["auto"], outs["f16x3"]) and not torch.equal(outs["auto"], outs["f32"])
    xs = x[..., :512].contiguous()
    for m in ("auto", "f32"):
        ops.hessian_set_method(m)
        h = torch.zeros((512, 512), device="cuda")
        ops.hessian_accumulate(xs, h, 0)
        outs[m] = h
    assert torch.equal(outs["auto"], outs["f32"])
    xt = x[:, :256].contiguous()                                     # same width, only 512 rows: below the measured cross-over
    for m in ("auto", "f32"):
        ops.hessian_set_method(m)
        h = torch.zeros((1024, 1024), device="cuda")
        ops.hessian_accumulate(xt, h, 0)
        outs[m] = h
    assert torch.equal(outs["auto"], outs["f32"])


def test_hessian_method_errors_are_loud(ops, restore_hessian_method):
    import torch
    from onnx_quantize_amd.hip import _lib as L
    lib = L.load()
    with pytest.raises(ValueError, match="unknown Hessian method"):
        ops.hessian_set_method("tf32")
    x = torch.randn((64, 256), device="cuda")
    h = torch.zeros((256, 256), device="cuda")
    ws = torch.empty(1024, dtype=torch.uint8, device="cuda")         # far too small for the pieces
    st = lib.oq_hessian_accumulate_f32(x.data_ptr(), 64, 256, 256, 0, 64, h.data_ptr(), 7, ws.data_ptr(), ws.numel(), None)
    assert st == L.OQ_ERR_INVALID_ARGUMENT and b"unknown method" in lib.oq_last_error()
    st = lib.oq_hessian_accumulate_f32(x.data_ptr(), 64, 256, 256, 0, 64, h.data_ptr(), ops.HESSIAN_METHODS["bf16x6"], ws.data_ptr(), ws.numel(), None)
    assert st == L.OQ_ERR_WORKSPACE and b"workspace" in lib.oq_last_error()
    assert float(h.abs().max()) == 0.0                               # nothing was written
    # auto never fails for lack of workspace: fp32 kernel
    st = lib.oq_hessian_accumulate_f32(x.data_ptr(), 64, 256, 256, 0, 64, h.data_ptr(), 0, None, 0, None)
    assert st == 0 and float(h.abs().max()) > 0


@pytest.mark.parametrize("k", [96, 128, 200, 256, 515, 1024])
def test_factor_vs_float64(ops, k):
    """U upper, zero below the diagonal, U^T U = inv(H + damp I)."""
    import torch
    rng = np.random.default_rng(k)
    x = rng.standard_normal((3 * k, k)).astype(np.float32) * rng.uniform(0.2, 2, size=k).astype(np.float32)
    h = ((2.0 / (3 * k)) * x.T @ x).astype(np.float32)
    u, info = ops.gptq_factor(dev(h), 0.01)
    assert int(info.cpu()) == 0
    u = u.cpu().numpy().astype(np.float64)
    assert np.all(np.tril(u, -1) == 0) and np.all(np.diag(u) > 0)
    hd = h.astype(np.float64)
    hd[np.diag_indices(k)] += 0.01 * np.mean(np.diag(h))
    inv = np.linalg.inv(hd)
    np.testing.assert_allclose(u.T @ u, inv, rtol=0, atol=2e-4 * np.abs(inv).max())
    # and against the oracle's factor (reference operation sequence)
    uo, ok = O.gptq_factor(h, 0.01)
    assert ok
    np.testing.assert_allclose(u, uo, rtol=0, atol=2e-4 * np.abs(uo).max())


def test_factor_not_spd_falls_back_to_identity(ops):
    import torch
    h = -np.eye(200, dtype=np.float32)
    u, info = ops.gptq_factor(dev(h), 0.01)
    assert int(info.cpu()) > 0
    np.testing.assert_array_equal(u.cpu().numpy(), np.eye(200, dtype=np.float32))


@pytest.mark.parametrize("k,count", [(96, 3), (200, 5), (515, 4), (1024, 3), (1536, 2)])
def test_batched_factor_is_the_single_factor_matrix_by_matrix(ops, k, count):
    """oq_gptq_factor_batched_f32: one chain of launches for `count` Hessians of one width.  Per matrix the operations are
    those of oq_gptq_factor_f32, so U and info are bit-identical -- including a member that is not positive definite
    (identity + info > 0, gptq.py:143-150) without disturbing its neighbours."""
    import torch
    rng = np.random.default_rng(k + count)
    hs = []
    for i in range(count):
        x = rng.standard_normal((3 * k, k)).astype(np.float32) * rng.uniform(0.2, 2, size=k).astype(np.float32)
        hs.append(((2.0 / (3 * k)) * x.T @ x).astype(np.float32))
    hs[1] = -hs[1]                                     # not SPD
    hb = dev(np.stack(hs))
    ub, ib = ops.gptq_factor_batched(hb, 0.01)
    assert ub.shape == (count, k, k) and ib.shape == (count,)
    for i in range(count):
        u1, i1 = ops.gptq_factor(dev(hs[i]), 0.01)
        assert torch.equal(ub[i], u1), i
        assert int(ib[i].cpu()) == int(i1.cpu()), i
    assert int(ib[1].cpu()) > 0 and int(ib[0].cpu()) == 0
    np.testing.assert_array_equal(ub[1].cpu().numpy(), np.eye(k, dtype=np.float32))


def test_batched_shared_factors_equal_the_per_input_ones(ops):
    """`gptq_shared_factors` (dead diagonal fixed inside the factor) against `gptq_shared_factor` per input, through a
    whole `gptq_quantize` call, with a dead channel in one of the inputs."""
    import torch
    w, x = GPTQ["b_w"], GPTQ["b_x"]
    k = w.shape[0]
    x2 = np.array(x, copy=True)
    x2[..., 5] = 0                                     # dead input channel
    h_a, _ = hessian_of(ops, x)
    h_b, _ = hessian_of(ops, x2)
    stack = torch.stack([h_a, h_b]).contiguous()
    batched = ops.gptq_shared_factors(stack, 0.01)
    for h, sh in zip((h_a, h_b), batched):
        single = ops.gptq_shared_factor(h, 0.01, False)
        assert torch.equal(sh["u"], single["u"]) and torch.equal(sh["dead"], single["dead"])
        assert int(sh["info"].cpu()) == int(single["info"].cpu())
        a = ops.gptq_quantize(dev(w), h, "int4", "group", 128)
        b = ops.gptq_quantize(dev(w), h, "int4", "group", 128, shared=sh)
        assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1]) and torch.equal(a[2], b[2])
    assert bool(batched[1]["dead"][5]) and not bool(batched[0]["dead"][5])
    assert int(batched[1]["dead"].sum()) == int(batched[0]["dead"].sum()) + 1      # the fixture has dead channels of its own


@pytest.mark.parametrize("case", GPTQ_CASES, ids=[c["id"] for c in GPTQ_CASES])
def test_gptq_parity_mode_vs_golden(ops, case):
    cid, d = case["id"], case["data"]
    w, x = GPTQ[f"{d}_w"], GPTQ[f"{d}_x"]
    h, _ = hessian_of(ops, x)
    q, s, z, info = ops.gptq_quantize(dev(w), h, case["qtype"], case["strategy"], case["group_size"],
                                      case["symmetric"], case["reduce_range"], case["clip_ratio"],
                                      case["block_size"], case["percdamp"], case["actorder"], case["mse"],
                                      mode="parity")
    gq, gs, gz = GPTQ[f"{cid}_q"], GPTQ[f"{cid}_s"], GPTQ[f"{cid}_z"]
    q, s, z = q.cpu().numpy(), s.cpu().numpy(), z.cpu().numpy()
    assert q.dtype == gq.dtype and q.shape == gq.shape
    np.testing.assert_array_equal(q, gq)
    assert z.dtype == gz.dtype and z.shape == gz.shape
    np.testing.assert_array_equal(z, gz)
    assert s.shape == gs.shape
    np.testing.assert_allclose(s, gs, rtol=1e-5, atol=0)


def test_gptq_corrected_mode_vs_oracle(ops):
    """The opt-in corrected update: integers may differ from the CPU oracle only where fp32 summation
    order flips a rounding; the layer output error must match the oracle's and beat RTN."""
    w, x = GPTQ["b_w"], GPTQ["b_x"]
    h, _ = hessian_of(ops, x)
    args = ("int4", "group", 128, False, False, 1.0, 128, 0.01, False, False)
    q, s, z, info = ops.gptq_quantize(dev(w), h, *args, mode="corrected")
    q, s, z = q.cpu().numpy(), s.cpu().numpy(), z.cpu().numpy()
    qo, so, zo = O.gptq_quantize(w, x, *args, mode="corrected")
    qp, sp, zp = O.gptq_quantize(w, x, *args, mode="parity")
    assert q.shape == qo.shape and s.shape == so.shape and z.shape == zo.shape
    mismatch = np.mean(q != qo)
    assert mismatch < 0.02, mismatch
    assert np.abs(q.astype(np.int32) - qo.astype(np.int32)).max() <= 1
    x2 = x.reshape(-1, x.shape[-1]).astype(np.float64)

    def out_err(qq, ss, zz):
        dq = O.dequantize(qq, ss, zz, rows_of="group", group_size=128).astype(np.float64)
        return np.linalg.norm(x2 @ dq - x2 @ w.astype(np.float64))
    e_gpu, e_cpu, e_rtn = out_err(q, s, z), out_err(qo, so, zo), out_err(qp, sp, zp)
    assert e_gpu < 0.9 * e_rtn
    assert abs(e_gpu - e_cpu) < 0.05 * e_cpu


def test_gptq_layer_shaped_property(ops):
    """Llama-shaped slice (K = 1024 -> N = 512), parity mode: integers == fused RTN kernel on the same
    device (the reference's documented behaviour), for group and channel strategies."""
    import torch
    gen = torch.Generator(device="cuda").manual_seed(5)
    k, n, t = 1024, 512, 2048
    w = torch.randn((k, n), generator=gen, device="cuda") * 0.02
    x = torch.randn((8, t // 8, k), generator=gen, device="cuda")
    h = torch.zeros((k, k), device="cuda")
    ops.hessian_accumulate(x, h, 0)
    for qtype, strategy, g in (("int4", "group", 128), ("int8", "channel", -1)):
        q, s, z, info = ops.gptq_quantize(w, h, qtype, strategy, g)
        assert int(info.cpu()) == 0
        rq, rs, rz = ops.rtn_quantize(w, qtype, strategy, g)
        assert torch.equal(q, rq) and torch.equal(z, rz)
        assert torch.allclose(s, rs, rtol=1e-5, atol=0)


def _mse_tables(rows, qtype, strategy, sym=False, red=False):
    """The oracle's error table [iterations, rows] of utils.py:140-239 on `rows` and the (scale, zp) of every candidate."""
    trace = []
    O.min_max_mse(rows, qtype, strategy, sym, red, trace=trace)
    errs = np.stack([np.asarray(e, dtype=np.float32).reshape(-1) for _, e in trace], axis=0)
    lo0, hi0 = O.min_max(rows, strategy, 1.0)
    cands = []
    for i in range(len(trace)):
        p = 1 - i / 100.0
        s, z = O.qparams(p * lo0, p * hi0, qtype, sym, red)
        cands.append((np.asarray(s, np.float32).reshape(-1), np.asarray(z).reshape(-1)))
    return errs, cands


@pytest.mark.parametrize("strategy,group_size,qtype", [("tensor", 8, "int8"), ("group", 8, "int4"), ("channel", -1, "int8"),
                                                       ("group", 16, "uint4"), ("tensor", 64, "int8")])
@pytest.mark.parametrize("actorder", [False, True])
def test_gptq_with_mse_vs_oracle(ops, strategy, group_size, qtype, actorder):
    """mse=True inside GPTQ (test_gptq.py:20-51 grid): the initial, per-group and final parameters all come from the MSE
    search, whose float32 pow / summation order cannot be reproduced bit for bit.  Same tolerance-aware contract as
    tests/test_mse_gpu.py, stage by stage:

    * integers: every block of rows that shares parameters (a loop group, gptq.py:168-184, else the initial per-channel /
      per-tensor ones, :104-116) must carry the integers of ONE of the reference's candidate ranges, and that candidate
      must be optimal for the oracle's own error table up to 1e-4;
    * returned (scale, zp): re-derived by the same search from the dequantized Q (:219-231) -- checked against the
      oracle's table on the Q the GPU itself produced, for every row whose Q is unambiguous;
    * and in practice nearly everything equals the oracle outright (>= 95 % of the integers, >= 90 % of the rows)."""
    w, x = GPTQ["a_w"], GPTQ["a_x"]
    k, n = w.shape
    h, _ = hessian_of(ops, x)
    q, s, z, info = ops.gptq_quantize(dev(w), h, qtype, strategy, group_size, False, False, 1.0, 128, 0.01, actorder, True)
    he, _ = O.accumulate_hessian(x, np.zeros((k, k), np.float32), 0)
    eq, es, ez, dbg = O.gptq(w, he, qtype, strategy, group_size, False, False, 1.0, 128, 0.01, actorder, True, return_debug=True)
    q, s, z = q.cpu().numpy(), s.cpu().numpy(), z.cpu().numpy()
    assert q.shape == eq.shape and s.shape == es.shape and z.shape == ez.shape and q.dtype == eq.dtype
    perm = dbg["perm"] if actorder else np.arange(k)
    wp = w.copy()
    wp[dbg["dead"], :] = 0
    wp, qp = wp[perm, :], q[perm, :]
    used = "channel" if strategy == "group" else strategy
    loop_g = group_size if (group_size and group_size != -1) else 0
    blocks = ([(r, min(r + loop_g, k), wp[r:r + loop_g, :].T, "channel") for r in range(0, k, loop_g)] if loop_g
              else [(0, k, w.T, used)])                                   # initial parameters see the untouched W (:104-116)
    qd = np.zeros_like(wp)
    unambiguous = np.ones((k, n), bool)
    for r0, r1, rows, st in blocks:
        errs, cands = _mse_tables(rows, qtype, st)
        best = errs.min(axis=0)
        data = wp[r0:r1, :].T if st == "channel" else wp[r0:r1, :].reshape(1, -1)
        got = qp[r0:r1, :].T if st == "channel" else qp[r0:r1, :].reshape(1, -1)
        nrows = data.shape[0]
        chosen = np.full(nrows, -1)
        matches = np.zeros(nrows, int)
        for i, (cs, cz) in enumerate(cands):
            ints = O.quantize(data, cs.reshape(-1, 1), cz.reshape(-1, 1), qtype, False, False)
            hit = np.all(ints == got, axis=1) & (errs[i] <= best * (1 + 1e-4) + 1e-30)
            matches += hit
            chosen[(chosen < 0) & hit] = i
        assert np.all(chosen >= 0), f"rows {r0}:{r1}: integers that belong to no near-optimal candidate of the reference's grid"
        cs = np.stack([c[0] for c in cands])[chosen, np.arange(nrows)].reshape(-1, 1)
        cz = np.stack([c[1] for c in cands])[chosen, np.arange(nrows)].reshape(-1, 1)
        deq = O.dequantize(got, cs, cz)
        if st == "channel":
            qd[r0:r1, :] = deq.T
            unambiguous[r0:r1, :] &= (matches == 1)[None, :]
        else:
            qd[r0:r1, :] = deq.reshape(r1 - r0, n)
            unambiguous[r0:r1, :] &= bool(matches[0] == 1)
    inv = np.argsort(perm)
    qd, unambiguous = qd[inv, :], unambiguous[inv, :]
    # final parameters: the search on the GPU's own dequantized Q in the user's layout
    rows = O.to_rows(qd, strategy, group_size)
    clear = np.all(O.to_rows(unambiguous, strategy, group_size), axis=-1).reshape(-1) if strategy != "tensor" else np.array([unambiguous.all()])
    errs, cands = _mse_tables(rows, qtype, strategy)
    sg, zg = s.reshape(-1), z.reshape(-1)
    idx = np.full(sg.size, -1)
    for i, (cs, cz) in enumerate(cands):
        hit = (idx < 0) & (cs.view(np.uint32) == sg.view(np.uint32)) & (cz == zg)
        idx[hit] = i
    ok = (idx >= 0) & (errs[np.maximum(idx, 0), np.arange(sg.size)] <= errs.min(axis=0) * (1 + 1e-4) + 1e-30)
    assert clear.mean() >= 0.9 and np.all(ok[clear]), f"{(~ok[clear]).sum()} returned (scale, zp) off the candidate grid / not optimal"
    assert np.mean(q == eq) >= 0.95
    same = (sg.view(np.uint32) == es.reshape(-1).view(np.uint32)) & (zg == ez.reshape(-1))
    assert same.mean() >= 0.9


@pytest.mark.parametrize("actorder", [False, True])
def test_shared_factor_equals_per_layer_factor(ops, actorder):
    """q/k/v (and gate/up) share their input: one Hessian + one inverse factor must give exactly the results of
    the reference's per-node recomputation."""
    import torch
    w, x = GPTQ["b_w"], GPTQ["b_x"]
    h, _ = hessian_of(ops, x)
    shared = ops.gptq_shared_factor(h, 0.01, actorder)
    for seed in (0, 1):
        wi = dev(w + np.random.default_rng(seed).standard_normal(w.shape).astype(np.float32) * 0.01)
        a = ops.gptq_quantize(wi, h, "int4", "group", 128, actorder=actorder)
        b = ops.gptq_quantize(wi, h, "int4", "group", 128, actorder=actorder, shared=shared)
        assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1]) and torch.equal(a[2], b[2])


def test_device_gather_single_process(ops):
    import torch
    from onnx_quantize_amd.sharding import LayerSpec, gather_device_results, plan_lpt
    specs = [LayerSpec("a", 64, 32), LayerSpec("b", 128, 16)]
    plan = plan_lpt(specs, 1)
    mine = {i: ops.rtn_quantize(torch.randn((s.k, s.n), device="cuda"), "uint4", "group", 32) for i, s in enumerate(specs)}
    out, nbytes = gather_device_results(specs, plan, mine)
    assert list(out) == ["a", "b"] and nbytes == 0 and out["b"][0].shape == (128, 16)


def test_full_size_layer_properties(ops):
    """Size-independent properties at a BASELINE layer size (K = N = 4096, Llama-2-7B q_proj; config 4), all on the
    device: (a) the Hessian of the streamed batches equals the float64 product; (b) U is upper triangular with
    U^T U (H + damp I) = I (checked in float64 through random probe vectors); (c) parity mode returns exactly the
    group-RTN integers (the reference's error feedback is a no-op, SURVEY.md finding 1) and the parameters
    re-derived from the dequantized matrix; (d) corrected mode lowers the layer-output error below RTN's."""
    import torch
    k = n = 4096
    gen = torch.Generator(device="cuda").manual_seed(11)
    chan = 0.1 + 3.9 * torch.rand(k, generator=gen, device="cuda")
    xs = [torch.randn((4, 1024, k), generator=gen, device="cuda") * chan for _ in range(2)]
    h = torch.zeros((k, k), device="cuda")
    cnt = 0
    for x in xs:
        cnt = ops.hessian_accumulate(x, h, cnt)
    x64 = torch.cat([x.reshape(-1, k) for x in xs]).double()
    h64 = (2.0 / cnt) * (x64.T @ x64)
    assert float((h.double() - h64).abs().max()) <= 2e-5 * float(h64.abs().max())
    u, info = ops.gptq_factor(h, 0.01)
    assert int(info.item()) == 0 and float(torch.tril(u, -1).abs().max()) == 0.0 and bool((torch.diagonal(u) > 0).all())
    hd = h.double() + 0.01 * torch.diagonal(h).double().mean() * torch.eye(k, device="cuda", dtype=torch.float64)
    probe = torch.randn((k, 16), generator=gen, device="cuda", dtype=torch.float64)
    back = u.double().T @ (u.double() @ (hd @ probe))
    assert float((back - probe).abs().max()) <= 5e-3 * float(probe.abs().max())
    w = torch.randn((k, n), generator=gen, device="cuda") * 0.02
    q, s, z, _ = ops.gptq_quantize(w, h, "int4", "group", 128)
    rq, rs, rz = ops.rtn_quantize(w, "int4", "group", 128)
    assert torch.equal(q, rq) and torch.equal(z, rz)
    torch.testing.assert_close(s, rs, rtol=1e-5, atol=0)
    qc, sc, zc, _ = ops.gptq_quantize(w, h, "int4", "group", 128, mode="corrected")
    def out_err(qq, ss, zz):
        dq = ops.dequantize(qq, ss, zz, "int4", mode="group", group=128)
        return float(((xs[0].reshape(-1, k)[:2048] @ (dq - w)) ** 2).mean())
    assert out_err(qc, sc, zc) < 0.9 * out_err(rq, rs, rz)


def test_down_proj_full_size_factor_and_parity_loop(ops):
    """BASELINE config 4's widest layer, Llama-2-7B down_proj: K = 11008, N = 4096 (VERDICT r01, weak spot: nothing under
    `-m gpu` factored K = 11008 or ran its loop).  (a) the inverse factor of the 11008 x 11008 Hessian: U upper, positive
    diagonal, U^T U (H + damp I) = I through random float64 probe vectors; (b) the parity loop over 86 blocks of 128 rows
    returns exactly the fused RTN kernel's integers and zero points (the reference's error feedback is a no-op,
    SURVEY.md finding 1), scales within the north-star tolerance; (c) the dead-channel rule (gptq.py:119-121) at this
    size: an input channel the calibration never excites quantizes a ZERO row, not the weight row."""
    import torch
    k, n = 11008, 4096
    gen = torch.Generator(device="cuda").manual_seed(21)
    chan = 0.1 + 3.9 * torch.rand(k, generator=gen, device="cuda")
    chan[4321] = 0
    h = torch.zeros((k, k), device="cuda")
    cnt = 0
    for _ in range(2):
        cnt = ops.hessian_accumulate(torch.randn((8, 1024, k), generator=gen, device="cuda") * chan, h, cnt)
    assert cnt == 16 and float(h[4321, 4321]) == 0.0
    h1 = h.clone()
    h1[4321, 4321] = 1.0                                               # what gptq.py:119-120 factors
    u, info = ops.gptq_factor(h1, 0.01)
    assert int(info.item()) == 0 and float(torch.tril(u, -1).abs().max()) == 0.0 and bool((torch.diagonal(u) > 0).all())
    damp = 0.01 * torch.diagonal(h1).double().mean()
    probe = torch.randn((k, 8), generator=gen, device="cuda", dtype=torch.float64)
    hp = h1.double() @ probe + damp * probe
    back = u.double().T @ (u.double() @ hp)
    assert float((back - probe).abs().max()) <= 5e-3 * float(probe.abs().max())
    del hp, back, u, h1
    w = torch.randn((k, n), generator=gen, device="cuda") * 0.02
    q, s, z, info = ops.gptq_quantize(w, h, "int4", "group", 128)
    assert int(info.item()) == 0
    w0 = w.clone()
    w0[4321, :] = 0                                                    # gptq.py:121
    rq, rs, rz = ops.rtn_quantize(w0, "int4", "group", 128)
    assert torch.equal(q, rq) and torch.equal(z, rz)
    torch.testing.assert_close(s, rs, rtol=1e-5, atol=0)
    pq, _, _ = ops.rtn_quantize(w, "int4", "group", 128)
    assert not torch.equal(q[4321], pq[4321])                          # the dead row really differs from plain RTN


def test_hessian_widest_llama_input_properties(ops):
    """K = 11008 (Llama-2-7B down_proj input, config 4): 43 x 43 tiles of the split-operand kernel, several T-slices.
    Exactly symmetric, the diagonal = (2/n) sum x^2 and a 256-column strip = the float64 product within the Hessian
    tolerance, a zero channel stays exactly zero, a second batch continues the running average."""
    import torch
    k = 11008
    gen = torch.Generator(device="cuda").manual_seed(5)
    chan = 0.1 + 3.9 * torch.rand(k, generator=gen, device="cuda")
    chan[777] = 0
    xs = [torch.randn((2, 1024, k), generator=gen, device="cuda") * chan + 0.1 * (chan > 0) for _ in range(2)]
    assert ops.hessian_method() == "auto"
    h = torch.zeros((k, k), device="cuda")
    n = 0
    for x in xs:
        n = ops.hessian_accumulate(x, h, n)
    assert n == 4 and torch.equal(h, h.T)
    x64 = torch.cat([x.reshape(-1, k) for x in xs]).double()
    top = float((2.0 / n) * (x64 * x64).sum(0).max())
    diag = (2.0 / n) * (x64 * x64).sum(0)
    assert float((torch.diagonal(h).double() - diag).abs().max()) <= 1e-5 * top
    strip = (2.0 / n) * (x64[:, 10752:].T @ x64)                     # the last, ragged-free tile column
    assert float((h[10752:].double() - strip).abs().max()) <= 1e-5 * top
    assert float(h[777].abs().max()) == 0.0 and float(h[:, 777].abs().max()) == 0.0


def test_results_are_deterministic(ops):
    """No atomics on floating point anywhere on the path: T-slices are summed in slice order, reductions fold in a fixed
    order -- the same call gives the same bits, for every Hessian kernel, the factor and both loop modes."""
    import torch
    gen = torch.Generator(device="cuda").manual_seed(3)
    x = torch.randn((4, 2048, 1280), generator=gen, device="cuda")
    w = torch.randn((1280, 768), generator=gen, device="cuda") * 0.05
    before = ops.hessian_method()
    try:
        for m in ("f32", "bf16x6", "bf16x9"):
            ops.hessian_set_method(m)
            hs = []
            for _ in range(2):
                h = torch.zeros((1280, 1280), device="cuda")
                n = ops.hessian_accumulate(x, h, 0)
                ops.hessian_accumulate(x, h, n)
                hs.append(h)
            assert torch.equal(hs[0], hs[1]), m
    finally:
        ops.hessian_set_method(before)
    u1, _ = ops.gptq_factor(hs[0], 0.01)
    u2, _ = ops.gptq_factor(hs[0], 0.01)
    assert torch.equal(u1, u2)
    for mode in ("parity", "corrected"):
        a = ops.gptq_quantize(w, hs[0], "int4", "group", 128, mode=mode)
        b = ops.gptq_quantize(w, hs[0], "int4", "group", 128, mode=mode)
        assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1]) and torch.equal(a[2], b[2]), mode


def test_calls_on_side_streams_do_not_interfere(ops):
    """Every entry point takes the stream it runs on and owns its workspace for the call: two streams working at the
    same time (a Hessian on one, RTN + calibration on the other) give what they give alone."""
    import torch
    gen = torch.Generator(device="cuda").manual_seed(9)
    x = torch.randn((4, 4096, 1024), generator=gen, device="cuda")
    w = torch.randn((2048, 1536), generator=gen, device="cuda")
    h_ref = torch.zeros((1024, 1024), device="cuda")
    ops.hessian_accumulate(x, h_ref, 0)
    q_ref, s_ref, z_ref = ops.rtn_quantize(w, "uint4", "group", 128, layout="nbits")
    torch.cuda.synchronize()
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    hs, qs = [], []
    for _ in range(4):
        with torch.cuda.stream(s1):
            h = torch.zeros((1024, 1024), device="cuda")
            ops.hessian_accumulate(x, h, 0)
            hs.append(h)
        with torch.cuda.stream(s2):
            qs.append(ops.rtn_quantize(w, "uint4", "group", 128, layout="nbits"))
    torch.cuda.synchronize()
    assert all(torch.equal(h, h_ref) for h in hs)
    assert all(torch.equal(q, q_ref) and torch.equal(s, s_ref) and torch.equal(z, z_ref) for q, s, z in qs)


def test_hessian_of_a_very_wide_input(ops):
    """K = 16384 (a 1 GiB Hessian, 2080 upper tiles): last-tile strip and diagonal against float64, exact symmetry."""
    import torch
    k = 16384
    gen = torch.Generator(device="cuda").manual_seed(2)
    x = torch.randn((2, 2048, k), generator=gen, device="cuda")
    h = torch.zeros((k, k), device="cuda")
    assert ops.hessian_accumulate(x, h, 0) == 2
    assert torch.equal(h, h.T)
    x64 = x.reshape(-1, k).double()
    strip = x64[:, k - 256:].T @ x64                                 # (2 / n) = 1
    top = float(strip.abs().max())
    assert float((h[k - 256:].double() - strip).abs().max()) <= 1e-5 * top
    assert float((torch.diagonal(h).double() - (x64 * x64).sum(0)).abs().max()) <= 1e-5 * top


def test_random_gptq_configurations_against_the_oracle(ops):
    """Property test (hypothesis, fixed seed) of the whole `_gptq_quantize` composition in parity mode: random shapes,
    types, strategies, group / block sizes, actorder, dead channels, symmetric / reduce_range / clip_ratio.  Integers and
    zero points equal the oracle's (which equals the reference on 104 golden cases), scales to 1e-5."""
    import torch
    from hypothesis import HealthCheck, given, seed, settings, strategies as st

    @st.composite
    def case(draw):
        qtype = draw(st.sampled_from(["int4", "uint4", "int8", "uint8"]))
        strategy = draw(st.sampled_from(["tensor", "channel", "group"]))
        k = draw(st.sampled_from([16, 32, 48, 64, 96, 128, 160, 256]))
        g = draw(st.sampled_from([d for d in (8, 16, 32, 64, 128) if k % d == 0])) if strategy == "group" else draw(st.sampled_from([32, -1]))
        n = draw(st.integers(1, 70))
        block = draw(st.sampled_from([16, 32, 64, 128]))
        return (qtype, strategy, k, g, n, block, draw(st.booleans()), draw(st.booleans()), draw(st.booleans()),
                draw(st.sampled_from([1.0, 0.9])), draw(st.integers(0, 3)), draw(st.integers(0, 2**31 - 1)))

    @seed(20240602)
    @settings(max_examples=60, deadline=None, suppress_health_check=list(HealthCheck))
    @given(case())
    def run(c):
        qtype, strategy, k, g, n, block, actorder, sym, red, clip, dead, rs = c
        r = np.random.default_rng(rs)
        w = (r.standard_normal((k, n)) * 0.2).astype(np.float32)
        x = (r.standard_normal((3, 24, k)) * r.uniform(0.2, 3.0, size=k)).astype(np.float32)
        for d in r.choice(k, size=min(dead, k), replace=False):
            x[..., d] = 0                                                   # dead input channels (gptq.py:118-123)
        eq, es, ez = O.gptq_quantize(w, x, qtype, strategy, g, sym, red, clip, block, 0.01, actorder)
        h = torch.zeros((k, k), device="cuda")
        ops.hessian_accumulate(dev(x), h, 0)
        q, s, z, _ = ops.gptq_quantize(dev(w), h, qtype, strategy, g, sym, red, clip, block, 0.01, actorder)
        np.testing.assert_array_equal(q.cpu().numpy(), eq, err_msg=str(c))
        np.testing.assert_array_equal(z.cpu().numpy().reshape(np.shape(ez)), ez, err_msg=str(c))
        np.testing.assert_allclose(s.cpu().numpy().reshape(np.shape(es)), es, rtol=1e-5, atol=0, err_msg=str(c))

    run()


def test_random_hessian_shapes_and_methods(ops, restore_hessian_method):
    """Property test (hypothesis, fixed seed): random T, K (not multiples of anything), batch counts, leading dimensions
    and kernels; every Hessian within 1e-5 max|H| of float64, exactly symmetric, counts right."""
    import torch
    from hypothesis import HealthCheck, given, seed, settings, strategies as st

    @seed(20240604)
    @settings(max_examples=40, deadline=None, suppress_health_check=list(HealthCheck))
    @given(st.sampled_from(["f32", "bf16x6", "bf16x9", "auto"]), st.integers(1, 700), st.integers(1, 9), st.integers(1, 300),
           st.integers(1, 3), st.integers(0, 5), st.integers(0, 2**31 - 1))
    def run(method, k, samples, seq, batches, pad, rs):
        ops.hessian_set_method(method)
        r = np.random.default_rng(rs)
        h = torch.zeros((k, k), device="cuda")
        n, xs = 0, []
        for _ in range(batches):
            x = (r.standard_normal((samples, seq, k + pad)) * r.uniform(0.1, 5)).astype(np.float32)
            xs.append(x[..., :k])
            n = ops.hessian_accumulate(dev(x)[..., :k], h, n)        # a strided view when pad > 0
        assert n == batches * samples
        x64 = np.concatenate([x.reshape(-1, k) for x in xs]).astype(np.float64)
        ref = (2.0 / n) * x64.T @ x64
        got = h.cpu().numpy()
        np.testing.assert_allclose(got, ref, rtol=0, atol=1e-5 * max(np.abs(ref).max(), 1e-30))
        np.testing.assert_array_equal(got, got.T)

    run()


# ---------------------------------------------------------------------------------------------------------- round 3
def _corr_inputs(k, n, t, seed):
    """Weights and CORRELATED calibration rows (a low-rank mix on top of per-channel noise), so that the error feedback of
    the corrected loop has something to do (independent channels give an almost diagonal Hessian)."""
    rng = np.random.default_rng(seed)
    w = (rng.standard_normal((k, n)) * 0.05).astype(np.float32)
    mix = rng.standard_normal((32, k)).astype(np.float32)
    x = (rng.standard_normal((t, 32)).astype(np.float32) @ mix + 0.5 * rng.standard_normal((t, k)).astype(np.float32))
    return w, x.reshape(8, t // 8, k)


_VARIANT_CASES = [  # (k, n, group, block): multiples of 16 rows (the rows-over-lanes kernel), ragged columns and last blocks
    (512, 200, 128, 128), (512, 200, 32, 128), (384, 68, 16, 64), (200, 52, None, 128), (512, 72, 64, 32), (1024, 130, 128, 128),
    (640, 64, 128, 256), (640, 64, 64, 320),
]


def _run_variant_cases(ops, mode="corrected"):
    import torch
    out = {}
    for ci, (k, n, g, bs) in enumerate(_VARIANT_CASES):
        w, x = _corr_inputs(k, n, 2048, 100 + ci)
        h = torch.zeros((k, k), device="cuda")
        ops.hessian_accumulate(torch.from_numpy(x).cuda(), h, 0)
        strategy = "group" if g else "channel"
        q, s, z, _ = ops.gptq_quantize(torch.from_numpy(w).cuda(), h, "int4", strategy, g if g else -1, block_size=bs, mode=mode)
        out[f"q{ci}"], out[f"s{ci}"], out[f"z{ci}"] = q.cpu().numpy(), s.cpu().numpy(), z.cpu().numpy()
    return out


def test_corrected_kernels_agree_bit_for_bit(ops):
    """The rows-over-lanes kernel of round 3 (16 lanes share a column, refined-reciprocal division, magic-number rounding)
    against the one-column-per-lane kernel it replaced (IEEE `/` everywhere): the same products, subtractions and quotients
    in the same order, so every integer, scale and zero point must be equal, bit for bit.  Since ABI 2 the old kernel is a
    mode of the call (OQ_GPTQ_CORRECTED_COLUMNS; until round 3 an environment variable read once per process)."""
    new = _run_variant_cases(ops)
    old = _run_variant_cases(ops, mode="corrected_columns")
    for key, val in new.items():
        assert val.dtype == old[key].dtype and np.array_equal(val, old[key]), key


@pytest.mark.parametrize("mode", ["parity", "corrected", "corrected_columns"])
@pytest.mark.parametrize("qtype", ["int4", "uint4"])
def test_gptq_loop_writes_the_packed_int4_layout_itself(ops, mode, qtype):
    """OQ_LAYOUT_KN_PACKED4 out of the three loop kernels (parity, rows-over-lanes, one column per lane): the bytes are
    core/_pack.py:8-22's serialisation of the [K, N] result of the same call (the separate
    `pack_nibbles` launch of the sharded bench is gone), scales and zero points unchanged; N = 132 leaves a ragged last
    workgroup."""
    import torch
    for (k, n, g) in ((512, 256, 128), (256, 132, 64), (384, 96, 48)):
        w, x = _corr_inputs(k, n, 1024, k + n)
        h = torch.zeros((k, k), device="cuda")
        ops.hessian_accumulate(torch.from_numpy(x).cuda(), h, 0)
        wd = torch.from_numpy(w).cuda()
        q, s, z, _ = ops.gptq_quantize(wd, h, qtype, "group", g, mode=mode)
        qp, sp, zp, _ = ops.gptq_quantize(wd, h, qtype, "group", g, mode=mode, layout="kn_packed4")
        assert qp.shape == (k, n // 2) and qp.dtype == torch.uint8
        assert qp.cpu().numpy().reshape(-1).tobytes() == O.pack_nibbles(q.cpu().numpy()).tobytes()
        assert torch.equal(s, sp) and torch.equal(z, zp)
    with pytest.raises(Exception, match="4-bit|packed"):
        ops.gptq_quantize(wd, h, "int8", "group", g, layout="kn_packed4")


@pytest.mark.parametrize("k,n,g,bs", [(512, 96, 64, 256), (640, 80, 32, 320), (512, 64, 128, 512), (384, 48, None, 256)])
def test_block_size_above_128_is_honoured(ops, k, n, g, bs):
    """blocks taller than 128 rows were silently walked as 128-row blocks.  Now the block is a chain of
    128-row launches on a working copy that alone receives the in-block updates, while group parameters are read from W as
    it stood when the block began (gptq.py:157, 168-184) -- which is what makes block_size matter when a group starts
    inside a block.  Against the oracle's corrected mode with the same block_size."""
    import torch
    w, x = _corr_inputs(k, n, 2048, k + n)
    h = torch.zeros((k, k), device="cuda")
    ops.hessian_accumulate(torch.from_numpy(x).cuda(), h, 0)
    strategy = "group" if g else "channel"
    args = ("int4", strategy, g if g else -1, False, False, 1.0)
    q, s, z, _ = ops.gptq_quantize(torch.from_numpy(w).cuda(), h, *args, block_size=bs, mode="corrected")
    qo, so, zo = O.gptq(w, h.cpu().numpy(), *args, bs, 0.01, False, False, mode="corrected")
    q, s, z = q.cpu().numpy(), s.cpu().numpy(), z.cpu().numpy()
    assert q.shape == qo.shape and s.shape == so.shape
    diff = np.abs(q.astype(np.int32) - qo.astype(np.int32))
    # a group whose range moved by one ulp can shift a zero point, and with it a level by two: rare, bounded
    assert np.mean(diff != 0) < 0.02 and np.mean(diff > 1) < 1e-3 and diff.max() <= 2
    np.testing.assert_allclose(s, so, rtol=2e-3)
    if g and g < bs:   # a group starts inside a block: its parameters see fewer updates than with 128-row blocks
        q128, _, _, _ = ops.gptq_quantize(torch.from_numpy(w).cuda(), h, *args, block_size=128, mode="corrected")
        q128 = q128.cpu().numpy()
        qo128, _, _ = O.gptq(w, h.cpu().numpy(), *args, 128, 0.01, False, False, mode="corrected")
        assert np.mean(qo != qo128) > 0.001                      # the oracle says block_size matters here ...
        assert np.mean(q != qo) < 0.5 * np.mean(q != qo128)      # ... and the GPU follows the block_size it was given


def test_corrected_mode_at_llama_size_follows_the_oracle_on_a_column_strip(ops):
    """nothing bounded the corrected mode at Llama sizes.  gate_proj shape, 4096 x 11008, int4 g128: output
    columns are independent given H, so the oracle's corrected mode on columns [0, 512) -- with the GPU's own Hessian --
    is the reference for that strip of the full-size GPU result: <= 2 % of the integers differ (fp32 summation order of
    the lazy updates and of the factor), almost all of them by one level, and the layer-output error is well below RTN's."""
    import torch
    k, n, strip = 4096, 11008, 512
    gen = torch.Generator(device="cuda").manual_seed(31)
    mix = torch.randn((64, k), generator=gen, device="cuda")
    x = torch.randn((8192, 64), generator=gen, device="cuda") @ mix + 0.5 * torch.randn((8192, k), generator=gen, device="cuda")
    w = torch.randn((k, n), generator=gen, device="cuda") * 0.02
    h = torch.zeros((k, k), device="cuda")
    ops.hessian_accumulate(x.reshape(8, 1024, k), h, 0)
    q, s, z, info = ops.gptq_quantize(w, h, "int4", "group", 128, mode="corrected")
    assert int(info.item()) == 0
    qo, so, zo = O.gptq(w[:, :strip].cpu().numpy(), h.cpu().numpy(), "int4", "group", 128, False, False, 1.0, 128, 0.01, False, False,
                        mode="corrected")
    qs = q[:, :strip].cpu().numpy()
    diff = np.abs(qs.astype(np.int32) - qo.astype(np.int32))
    mismatch = float(np.mean(diff != 0))
    assert mismatch < 0.02, mismatch
    # one level where a rounding flipped; more only where a flipped rounding early in a column was fed back into its later rows
    assert float(np.mean(diff > 1)) < 2e-3 and diff.max() <= 4, (float(np.mean(diff > 1)), int(diff.max()))
    # ... and that claim is checked, not assumed (the bar was loosened from 1 level to 4 after this test went
    # red, with the cause only asserted in a comment).  A single flipped rounding moves an integer by exactly one level, so
    # the FIRST difference of every column must be one level; anything larger has to lie below it (later rows of the same
    # column, which received the flipped row's error).  A first difference of two or more levels would be an indexing error
    # in gptq_rows16_kernel / panel_update_kernel / the deferred GEMM, not summation order.
    rows_first = np.argmax(diff != 0, axis=0)
    first = np.where(diff.any(axis=0), diff[rows_first, np.arange(strip)], 0)
    assert int(first.max()) <= 1, (int(first.max()), int(np.sum(first > 1)), rows_first[first > 1][:8].tolist())
    big_r, big_c = np.nonzero(diff > 1)
    assert np.all(big_r > rows_first[big_c]), "a difference of more than one level above its column's first difference"
    groups = k // 128
    # a group's scale hangs on its two extreme elements: where the feedback moved one of them differently the scale follows
    rel = np.abs(s.reshape(n, groups)[:strip].cpu().numpy() - so.reshape(strip, groups)) / so.reshape(strip, groups)
    assert float(np.mean(rel > 5e-3)) < 0.03 and float(rel.max()) < 0.1, (float(np.mean(rel > 5e-3)), float(rel.max()))
    rq, rs, rz = ops.rtn_quantize(w, "int4", "group", 128)
    x64 = x[:2048].double()

    def out_err(qq, ss, zz):
        dq = ops.dequantize(qq, ss, zz, "int4", mode="group", group=128)
        return float(torch.linalg.norm(x64 @ (dq - w).double()))
    e_gptq, e_rtn = out_err(q, s, z), out_err(rq, rs, rz)
    assert e_gptq < 0.8 * e_rtn, (e_gptq, e_rtn)


def test_two_stream_hessian_pipeline_gives_the_one_call_bits(ops):
    """oq_hessian_prepare_f32 + oq_hessian_accumulate_prepared_f32 through `ops.HessianPipeline` (preparation of the next
    batch on a side stream beside the product of this one) against `hessian_accumulate` with the fp16-piece method, over
    two inputs of different widths and batch counts: bit-identical Hessians, same sample counts."""
    import torch
    before = ops.hessian_method()
    ops.hessian_set_method("f16x3")
    try:
        gen = torch.Generator(device="cuda").manual_seed(41)
        inputs = {1024: [torch.randn((3, 700, 1024), generator=gen, device="cuda") * 3 for _ in range(3)],
                  2304: [torch.randn((2, 1000, 2304), generator=gen, device="cuda") for _ in range(2)] + [torch.randn((1, 1000, 2304), generator=gen, device="cuda")]}
        ref = {}
        for k, batches in inputs.items():
            h = torch.zeros((k, k), device="cuda")
            n = 0
            for x in batches:
                n = ops.hessian_accumulate(x, h, n)
            ref[k] = (h, n)
        pipe = ops.HessianPipeline(torch.device("cuda", torch.cuda.current_device()))
        seq = [(k, i) for k in inputs for i in range(len(inputs[k]))]
        got = {k: torch.zeros((k, k), device="cuda") for k in inputs}
        n = 0
        for si, (k, i) in enumerate(seq):
            if i == 0:
                n = 0
            x = inputs[k][i]
            nxt, nxt_total = None, None
            if si + 1 < len(seq):
                k2, i2 = seq[si + 1]
                nxt = inputs[k2][i2]
                nxt_total = (n + x.shape[0] if i2 != 0 else 0) + nxt.shape[0]
            n = pipe.accumulate(x, got[k], n, nxt, nxt_total)
            if i == len(inputs[k]) - 1:
                assert n == ref[k][1]
        torch.cuda.synchronize()
        for k in inputs:
            assert torch.equal(got[k], ref[k][0]), k
        # without look-ahead (next_x = None) the pipeline prepares on demand: same bits
        h2 = torch.zeros((1024, 1024), device="cuda")
        n = 0
        for x in inputs[1024]:
            n = pipe.accumulate(x, h2, n)
        assert torch.equal(h2, ref[1024][0])
    finally:
        ops.hessian_set_method(before)


def test_hessian_pipeline_never_reuses_stale_pieces(ops):
    """the pipeline recognised "prepared ahead" by (data_ptr, shape, n_total).  A batch edited in place after
    the look-ahead, or a NEW batch the allocator put at the address of a dropped one, matched that tag and the product ran
    on the stale fp16 pieces.  The tag now holds the tensor (its address cannot be recycled while held) and its version."""
    import torch
    before = ops.hessian_method()
    ops.hessian_set_method("f16x3")
    try:
        gen = torch.Generator(device="cuda").manual_seed(43)
        k = 1024
        a = torch.randn((2, 600, k), generator=gen, device="cuda")
        b = torch.randn((2, 600, k), generator=gen, device="cuda")
        pipe = ops.HessianPipeline(torch.device("cuda", torch.cuda.current_device()))
        h = torch.zeros((k, k), device="cuda")
        n = pipe.accumulate(a, h, 0, b, 4)           # b's pieces are prepared ahead ...
        b.mul_(3.0)                                  # ... and then b is edited in place
        n = pipe.accumulate(b, h, n)
        ref = torch.zeros((k, k), device="cuda")
        m = ops.hessian_accumulate(a, ref, 0)
        m = ops.hessian_accumulate(b, ref, m)
        assert n == m == 4 and torch.equal(h, ref)
        # a dropped look-ahead batch whose memory the allocator hands to the next one
        h2 = torch.zeros((k, k), device="cuda")
        c = torch.randn((2, 600, k), generator=gen, device="cuda")
        n = pipe.accumulate(a, h2, 0, c, 4)
        del c
        d = torch.randn((2, 600, k), generator=gen, device="cuda") * 0.5     # typically lands at c's address
        n = pipe.accumulate(d, h2, n)
        ref2 = torch.zeros((k, k), device="cuda")
        m = ops.hessian_accumulate(a, ref2, 0)
        m = ops.hessian_accumulate(d, ref2, m)
        assert torch.equal(h2, ref2)
    finally:
        ops.hessian_set_method(before)


# ----------------------------------------------------------------------------- a calibration batch's Hessians in one call
def _f64_update(h0, n_seen, x):
    """gptq.py:246-260 in float64 on the flattened input (n = leading-dimension entries, :247)."""
    n_add = x.shape[0]
    x2 = x.reshape(-1, x.shape[-1]).astype(np.float64)
    n = n_seen + n_add
    return h0.astype(np.float64) * (n_seen / n) + (2.0 / n) * (x2.T @ x2), n


def test_hessians_of_a_batch_in_one_call_follow_float64_and_the_per_tensor_route(ops):
    """`ops.hessian_accumulate_many` (oq_hessian_accumulate_many_f32): mixed widths and lengths -- gemma-3-270m's 640 / 1024 /
    2048 columns by 5120 rows, ragged ones, one below the grouped route's limits --, some Hessians already holding samples.
    Every result within 1e-5 max|H| of float64 (the north-star tolerance) and of the per-tensor call, exactly symmetric, dead
    channels exact zeros."""
    import torch
    rng = np.random.default_rng(11)
    shapes = [((10, 512, 640), 0), ((10, 512, 1024), 30), ((10, 512, 2048), 0), ((3, 333, 700), 7), ((1, 512, 513), 0),
              ((4, 200, 640), 12), ((2, 64, 96), 5), ((10, 512, 640), 10)]
    xs, hs, seen, refs, singles = [], [], [], [], []
    for shape, n_seen in shapes:
        k = shape[-1]
        x = rng.standard_normal(shape, dtype=np.float32) * rng.uniform(0.05, 4.0, size=k).astype(np.float32)
        x[..., 3] = 0.0                                               # a dead channel
        x[..., 5] *= 60.0                                             # an outlier channel
        a = rng.standard_normal((k + 8, k)).astype(np.float32)
        h0 = ((a.T @ a) / (k + 8)).astype(np.float32) if n_seen else np.zeros((k, k), np.float32)
        h0[3, :] = 0.0
        h0[:, 3] = 0.0
        ref, _ = _f64_update(h0, n_seen, x)
        h_single = dev(h0)
        ops.hessian_accumulate(dev(x), h_single, n_seen)
        xs.append(dev(x)); hs.append(dev(h0)); seen.append(n_seen); refs.append(ref); singles.append(h_single)
    out = ops.hessian_accumulate_many(xs, hs, seen)
    assert out == [s + sh[0][0] for s, sh in zip(seen, shapes)]
    for h, ref, single, (shape, _) in zip(hs, refs, singles, shapes):
        got = h.cpu().numpy()
        tol = 1e-5 * float(np.abs(ref).max())
        np.testing.assert_allclose(got, ref, rtol=0, atol=tol, err_msg=str(shape))
        np.testing.assert_allclose(got, single.cpu().numpy(), rtol=0, atol=tol, err_msg=str(shape))
        np.testing.assert_array_equal(got, got.T)
        assert got[3, 3] == 0.0 and not got[3].any() and not got[:, 3].any()
    # the item below the limits took the per-tensor route: the same bits
    assert torch.equal(hs[6], singles[6])


def test_hessians_of_a_batch_stream_like_the_per_tensor_accumulators(ops):
    """Three batches through the grouped call == the running update of gptq.py:254-258 in float64, and a repeated run gives
    the same bits (one fixed summation order per item)."""
    import torch
    rng = np.random.default_rng(12)
    widths = [640, 1024, 2048, 640]
    batches = [[rng.standard_normal((5, 256, k), dtype=np.float32) * 1.5 for k in widths] for _ in range(3)]

    def run():
        hs = [torch.zeros((k, k), device="cuda") for k in widths]
        n = [0] * len(widths)
        for b in batches:
            n = ops.hessian_accumulate_many([dev(x) for x in b], hs, n)
        return hs, n

    hs, n = run()
    hs2, _ = run()
    assert n == [15] * len(widths)
    for i, k in enumerate(widths):
        ref, seen = np.zeros((k, k)), 0
        for b in batches:
            ref, seen = _f64_update(ref, seen, b[i])
        np.testing.assert_allclose(hs[i].cpu().numpy(), ref, rtol=0, atol=1e-5 * float(np.abs(ref).max()))
        assert torch.equal(hs[i], hs2[i])


def test_hessians_of_a_batch_honour_the_method_knob_and_reject_bad_items(ops):
    import torch
    from onnx_quantize_amd.hip import _lib as L
    rng = np.random.default_rng(13)
    x = dev(rng.standard_normal((4, 256, 640), dtype=np.float32))
    before = ops.hessian_method()
    ops.hessian_set_method("f32")
    try:
        a, b = torch.zeros((640, 640), device="cuda"), torch.zeros((640, 640), device="cuda")
        ops.hessian_accumulate_many([x], [a], [0])                     # per-tensor route on the fp32 kernel
        ops.hessian_accumulate(x, b, 0)
        assert torch.equal(a, b)
        import ctypes as C
        lib = L.load()
        host = np.asarray([[x.data_ptr(), a.data_ptr(), 1024, 640, 640, 0, 4, 0]], dtype=np.int64)
        d = torch.from_numpy(host).cuda()
        ws = torch.empty(lib.oq_hessian_many_workspace_bytes(C.c_void_p(host.ctypes.data), 1), dtype=torch.uint8, device="cuda")
        # ABI 2: the grouped entry point IS the fp16-piece method -- it consults no process-wide setting any more (the thread's
        # default above only routes `ops.hessian_accumulate_many` in Python); called directly it runs and stays within the bound
        a.zero_()
        st = lib.oq_hessian_accumulate_many_f32(C.c_void_p(host.ctypes.data), C.c_void_p(d.data_ptr()), 1, C.c_void_p(ws.data_ptr()), ws.numel(), None)
        assert st == 0
        assert float((a - b).abs().max()) <= 1e-5 * float(b.abs().max()) and not torch.equal(a, b)
    finally:
        ops.hessian_set_method(before)
    with pytest.raises(ValueError):
        ops.hessian_accumulate_many([x], [torch.zeros((512, 512), device="cuda")], [0])
    host = np.asarray([[x.data_ptr(), a.data_ptr(), 1024, 640, 600, 0, 4, 0]], dtype=np.int64)      # ldx < K
    import ctypes as C
    lib = L.load()
    d = torch.from_numpy(host).cuda()
    ws = torch.empty(1 << 24, dtype=torch.uint8, device="cuda")
    assert lib.oq_hessian_accumulate_many_f32(C.c_void_p(host.ctypes.data), C.c_void_p(d.data_ptr()), 1, C.c_void_p(ws.data_ptr()), ws.numel(), None) == L.OQ_ERR_INVALID_ARGUMENT
    assert lib.oq_hessian_accumulate_many_f32(C.c_void_p(host.ctypes.data), C.c_void_p(d.data_ptr()), 1, C.c_void_p(ws.data_ptr()), 1024, None) != 0


# ----------------------------------------------------------------------------- the factor's large products on fp16 pieces
@pytest.mark.parametrize("k", [2601, 4224])
def test_factor_with_piece_products_follows_float64_and_the_fp32_kernel(ops, k):
    """From 1024-column trailing squares / 2048-row inverse blocks on, the factor's products take 22-bit operands on the fp16
    matrix cores (factor.hip).  Widths that exercise those branches with ragged edges (2601: not a multiple of 4, scalar
    epilogue; 4224 = 2 x 2048 + 128: a full pair and a ragged pair on the top level): U against a float64 factorisation
    (on the device, checker only), against the same call with fp32 products, and batched == single, bit for bit."""
    import torch
    g = torch.Generator(device="cuda").manual_seed(k)
    x = torch.randn((3 * k, k), generator=g, device="cuda") * (0.2 + 1.8 * torch.rand(k, generator=g, device="cuda"))
    h = torch.zeros((k, k), device="cuda")
    ops.hessian_accumulate(x.reshape(3, k, k), h, 0)
    hd = h.double()
    hd = hd + 0.01 * hd.diagonal().mean() * torch.eye(k, device="cuda", dtype=torch.float64)
    ref = torch.linalg.cholesky(torch.linalg.inv(hd)).T
    before = ops.hessian_method()
    try:
        got = {}
        for method in ("auto", "f32"):
            ops.hessian_set_method(method)
            u, info = ops.gptq_factor(h, 0.01)
            assert int(info.item()) == 0
            assert float(torch.tril(u, -1).abs().max()) == 0.0 and bool((u.diagonal() > 0).all())
            got[method] = u
            err = float((u.double() - ref).abs().max() / ref.abs().max())
            assert err < 2e-5, (method, err)
        assert float((got["auto"] - got["f32"]).abs().max() / ref.abs().max()) < 2e-5
        assert not torch.equal(got["auto"], got["f32"])             # the piece kernels really ran
        ops.hessian_set_method("auto")
        ub, ib = ops.gptq_factor_batched(torch.stack([h, 2.0 * h, h]), 0.01)
        assert torch.equal(ub[0], got["auto"]) and torch.equal(ub[2], got["auto"]) and ib.tolist() == [0, 0, 0]
    finally:
        ops.hessian_set_method(before)


def test_hessians_of_a_batch_property(ops):
    """Property test (hypothesis, fixed seed) of the grouped call: 1-5 items per call with widths that are multiples of
    nothing, lengths from 512 to 3000 rows, padded leading dimensions, Hessians that already hold samples -- every result
    within 1e-5 max|H| of float64 and exactly symmetric (checker on the device in float64)."""
    import torch
    from hypothesis import HealthCheck, given, seed, settings, strategies as st

    item = st.tuples(st.integers(512, 1500), st.integers(512, 3000), st.integers(0, 7), st.integers(0, 40), st.integers(1, 4))

    @seed(20261004)
    @settings(max_examples=25, deadline=None, suppress_health_check=list(HealthCheck))
    @given(st.lists(item, min_size=1, max_size=5), st.integers(0, 2**31 - 1))
    def run(items, sd):
        g = torch.Generator(device="cuda").manual_seed(sd)
        xs, hs, seen, refs = [], [], [], []
        for k, t, pad, n_seen, lead in items:
            t = t // lead * lead
            full = torch.randn((t, k + pad), generator=g, device="cuda") * (0.05 + 3.0 * torch.rand(k + pad, generator=g, device="cuda"))
            x = full[:, :k].reshape(lead, t // lead, k) if pad == 0 else full[:, :k].unflatten(0, (lead, t // lead))
            a = torch.randn((k, k), generator=g, device="cuda")
            h0 = (a @ a.T / k) if n_seen else torch.zeros((k, k), device="cuda")
            h0 = torch.triu(h0) + torch.triu(h0, 1).T           # exactly symmetric, as every H the path produces is
            n = n_seen + lead
            x2 = full[:, :k].double()
            refs.append(h0.double() * (n_seen / n) + (2.0 / n) * (x2.T @ x2))
            xs.append(x); hs.append(h0.clone()); seen.append(n_seen)
        out = ops.hessian_accumulate_many(xs, hs, seen)
        assert out == [s + it[4] for s, it in zip(seen, items)]
        for h, ref in zip(hs, refs):
            assert float((h.double() - ref).abs().max()) <= 1e-5 * float(ref.abs().max())
            assert torch.equal(h, h.T)

    run()
