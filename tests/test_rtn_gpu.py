"""GPU parity of the RTN path (A1 and its parts) through the C ABI: bit-exact integers and
zero points, fp32 scales within 1e-5 relative (in practice bit-equal) -- the tolerance the
north star states."""
import hashlib

import numpy as np
import pytest

import oq_oracle as O
from conftest import load_json, load_npz, synth_weight

pytestmark = pytest.mark.gpu

SCALE_RTOL = 1e-5


def sha16(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()[:16]


@pytest.fixture(scope="module")
def ops():
    import torch
    from onnx_quantize_amd.hip import ops as _ops
    assert torch.cuda.is_available()
    return _ops


def dev(a):
    import torch
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def run_rtn(ops, w, case, **kw):
    q, s, z = ops.rtn_quantize(dev(w), case["qtype"], case["strategy"], case["group_size"],
                               case["symmetric"], case["reduce_range"], case["clip_ratio"],
                               case.get("mse", False), **kw)
    return (None if q is None else q.cpu().numpy()), s.cpu().numpy(), z.cpu().numpy()


RTN_CASES = load_json("rtn_small.json")
RTN = load_npz("rtn_small.npz")


@pytest.mark.parametrize("case", RTN_CASES, ids=[c["id"] for c in RTN_CASES])
def test_rtn_small_vs_golden(ops, case):
    cid = case["id"]
    q, s, z = run_rtn(ops, RTN[f"{cid}_w"], case)
    gq, gs, gz = RTN[f"{cid}_q"], RTN[f"{cid}_s"], RTN[f"{cid}_z"]
    assert q.dtype == gq.dtype and q.shape == gq.shape
    np.testing.assert_array_equal(q, gq)
    assert z.dtype == gz.dtype and z.shape == gz.shape
    np.testing.assert_array_equal(z, gz)
    assert s.dtype == np.float32 and s.shape == gs.shape
    np.testing.assert_allclose(s, gs, rtol=SCALE_RTOL, atol=0)
    assert s.tobytes() == gs.tobytes()       # stronger than required: same bits


@pytest.mark.parametrize("case", RTN_CASES[::7], ids=[c["id"] for c in RTN_CASES[::7]])
def test_rtn_qparams_only(ops, case):
    cid = case["id"]
    q, s, z = run_rtn(ops, RTN[f"{cid}_w"], case, emit_q=False)
    assert q is None
    assert s.tobytes() == RTN[f"{cid}_s"].tobytes()
    np.testing.assert_array_equal(z, RTN[f"{cid}_z"])


@pytest.mark.parametrize("shape", [(128, 1024), (256, 260), (512, 255), (96, 1), (130, 2052)])
@pytest.mark.parametrize("g", [16, 32, 64, 128, 2, 48])
def test_rtn_group_ragged_columns_vs_oracle(ops, shape, g):
    """Column tails (N % 256, N % 4) and non-contiguous leading dimension."""
    k, n = shape
    if k % g:
        k = (k // g + 1) * g
    rng = np.random.default_rng(k * 131 + n + g)
    big = rng.standard_normal((k, n + 12), dtype=np.float32)
    import torch
    wd = dev(big)[:, 4:4 + n]                                  # ldw = n + 12, base offset 16 B
    w = big[:, 4:4 + n]
    for qtype, sym in (("uint4", False), ("int8", True)):
        q, s, z = ops.rtn_quantize(wd, qtype, "group", g, sym)
        eq, es, ez = O.rtn_quantize(w, qtype, "group", g, sym)
        np.testing.assert_array_equal(q.cpu().numpy(), eq)
        np.testing.assert_array_equal(z.cpu().numpy(), ez)
        assert s.cpu().numpy().tobytes() == es.tobytes()
    wd2 = dev(big)[:, 1:1 + n]                                  # misaligned base -> scalar path
    q, s, z = ops.rtn_quantize(wd2, "uint4", "group", g, False)
    eq, es, ez = O.rtn_quantize(big[:, 1:1 + n], "uint4", "group", g, False)
    np.testing.assert_array_equal(q.cpu().numpy(), eq)
    assert s.cpu().numpy().tobytes() == es.tobytes()
    np.testing.assert_array_equal(z.cpu().numpy(), ez)


@pytest.mark.parametrize("strategy,g", [("channel", -1), ("tensor", -1), ("group", 512), ("group", 384),
                                        ("group", -1)])
def test_rtn_two_pass_vs_oracle(ops, strategy, g):
    rng = np.random.default_rng(9)
    w = rng.standard_t(3, size=(1536, 516)).astype(np.float32)
    for qtype, sym, red in (("int8", False, False), ("uint8", True, False), ("int4", True, True),
                            ("uint4", False, False)):
        q, s, z = ops.rtn_quantize(dev(w), qtype, strategy, g, sym, red, 0.95)
        eq, es, ez = O.rtn_quantize(w, qtype, strategy, g, sym, red, 0.95)
        np.testing.assert_array_equal(q.cpu().numpy(), eq)
        np.testing.assert_array_equal(z.cpu().numpy(), ez)
        assert s.cpu().numpy().tobytes() == np.asarray(es).tobytes()
        assert s.shape == es.shape and z.shape == ez.shape


@pytest.mark.parametrize("k,n,strategy,g", [(4100, 260, "channel", -1), (8192, 516, "channel", -1), (12288, 300, "channel", -1),
                                            (16384, 264, "group", 8192), (24576, 68, "channel", -1), (9000, 1028, "group", 4500),
                                            (4000, 9000, "channel", -1), (4000, 9000, "group", 2000)])
def test_streamed_kernel_for_tall_ranges_vs_oracle(ops, k, n, strategy, g):
    """Ranges taller than 4096 rows -- or calls with at least four 128-row tiles per workgroup -- run on `rtn_resident_stream`
    (persistent workgroups, two tile slots, the next tile loaded while the previous one is stored): ragged last chunks,
    partial column tiles, two k-groups, up to the 192-chunk limit, short ranges over many columns; the
    state must come back zero (ops.rtn_quantize reuses one state buffer per stream without clearing it)."""
    import torch
    rng = np.random.default_rng(k + n)
    w = (rng.standard_t(4, size=(k, n)) * 0.05).astype(np.float32)
    wd = dev(w)
    for qtype, sym in (("int8", False), ("uint8", True), ("uint4", False)):
        for _ in range(2):                                    # the second call meets the state the first one left
            q, s, z = ops.rtn_quantize(wd, qtype, strategy, g, sym)
        eq, es, ez = O.rtn_quantize(w, qtype, strategy, g, sym)
        np.testing.assert_array_equal(q.cpu().numpy(), eq)
        np.testing.assert_array_equal(z.cpu().numpy(), ez)
        assert s.cpu().numpy().tobytes() == np.asarray(es).tobytes()
    state = ops._rtn_state(1, wd.device)                    # the buffer the calls above used (same device, same stream)
    torch.cuda.synchronize()
    assert int(state.count_nonzero()) == 0


DIGESTS = load_json("digests.json")


def test_config1_plumbing(ops):
    """BASELINE.json configs[0]: 256x512, QInt8 symmetric per-tensor."""
    d = DIGESTS["config1"]
    w = synth_weight(d["kind"], d["seed"], d["k"], d["n"])
    q, s, z = ops.rtn_quantize(dev(w), d["qtype"], d["strategy"], d["group_size"], d["symmetric"])
    assert sha16(q.cpu().numpy()) == d["q_sha"]
    assert s.cpu().numpy().tobytes().hex() == d["scale_hex"] and int(z.cpu()) == d["zp"]


@pytest.mark.parametrize("key", ["config2_asym", "config2_sym", "config2_heavy", "config2_zero_groups",
                                 "channel_4096", "int4_g128_4096", "headline_int8_channel", "headline_int8_tensor",
                                 "tall_int8_channel", "tall_int8_tensor"])
def test_full_size_digests(ops, key):
    """BASELINE.json configs[1] (4096x11008 uint4 g128) and friends at full size, against digests
    of what the reference itself produced (tests/golden/make_golden.py::gen_digests)."""
    d = DIGESTS[key]
    w = synth_weight(d["kind"], d["seed"], d["k"], d["n"])
    assert sha16(w) == d["w_sha"]
    q, s, z = ops.rtn_quantize(dev(w), d["qtype"], d["strategy"], d["group_size"], d["symmetric"])
    assert sha16(q.cpu().numpy()) == d["q_sha"]
    assert sha16(s.cpu().numpy()) == d["s_sha"]
    assert sha16(z.cpu().numpy()) == d["z_sha"]


def test_nbits_layout_matches_reference_packing(ops):
    """OQ_LAYOUT_NBITS == qrules/_common.py:65-123 applied to the [K, N] result."""
    rng = np.random.default_rng(17)
    # g in {32, 64, 128}: wave-owns-group kernel (ragged right edges: n % 32, n % 64, n % 128 != 0); other g: block kernel
    for (k, n, g, qtype) in [(512, 520, 128, "uint4"), (256, 36, 16, "uint4"), (512, 260, 64, "uint8"),
                             (768, 256, 256, "uint4"), (160, 40, 32, "uint8"), (384, 1100, 128, "uint8"),
                             (1024, 172, 64, "uint4"), (96, 2052, 32, "uint4"), (128, 4, 128, "uint4"),
                             (4096, 96, 128, "uint4"), (512, 72, 512, "uint4"), (1024, 44, -1, "uint8"), (384, 256, 384, "uint4")]:
        g = k if g == -1 else g            # group_size = -1 (whole column) is MatMulNBits-eligible (qrules/_common.py:32-62): two-pass path
        w = rng.standard_normal((k, n), dtype=np.float32)
        eq, es, ez = O.rtn_quantize(w, qtype, "group", g)
        eb, es2, epz = O.matmul_nbits_layout(eq, es, ez, g, O.BITWIDTH[qtype])
        b, s, z = ops.rtn_quantize(dev(w), qtype, "group", g, layout="nbits")
        np.testing.assert_array_equal(b.cpu().numpy(), eb)
        assert s.cpu().numpy().reshape(n, k // g).tobytes() == es2.tobytes()
        np.testing.assert_array_equal(z.cpu().numpy().reshape(-1), ez.reshape(-1))
        if qtype == "uint4" and k // g > 1:
            pz = ops.pack_zero_points_u4(z, n, k // g)
            np.testing.assert_array_equal(pz.cpu().numpy(), epz)
        # the same matrix as a column slice of a wider one (leading dimension > N, 16-byte aligned start)
        wide = np.zeros((k, n + 24), np.float32)
        wide[:, 8:8 + n] = w
        b2, s2, z2 = ops.rtn_quantize(dev(wide)[:, 8:8 + n], qtype, "group", g, layout="nbits")
        np.testing.assert_array_equal(b2.cpu().numpy(), eb)
        assert s2.cpu().numpy().tobytes() == s.cpu().numpy().tobytes()


@pytest.mark.parametrize("k,n", [(640, 4096), (640, 4224), (384, 8192), (256, 16384), (1152, 2176), (640, 12288)])
def test_block_orders_of_wide_matrices_vs_oracle(ops, k, n):
    """The id -> tile maps of the group kernels are chosen from the width (round 5: an XCD owns 2 / 8 / 1 neighbouring column
    tiles of a chunk for the blob kernel -- N = 4096, 4224: two, with a narrower last chunk; 8192: eight; 16384: one --, and
    the [K,N] layouts of 16 column tiles (N = 4096) run on the chunked order as well).  Every map has to visit every tile
    exactly once whatever the band count (640 rows = 5 k-groups: a short last band of the 4- and 8-group bands): all three
    layouts against the oracle."""
    rng = np.random.default_rng(k + n)
    w = rng.standard_normal((k, n), dtype=np.float32)
    eq, es, ez = O.rtn_quantize(w, "uint4", "group", 128)
    eb, es2, _ = O.matmul_nbits_layout(eq, es, ez, 128, 4)
    wd = dev(w)
    b, s, z = ops.rtn_quantize(wd, "uint4", "group", 128, layout="nbits")
    np.testing.assert_array_equal(b.cpu().numpy(), eb)
    assert s.cpu().numpy().reshape(n, k // 128).tobytes() == es2.tobytes()
    np.testing.assert_array_equal(z.cpu().numpy().reshape(-1), ez.reshape(-1))
    q, s, z = ops.rtn_quantize(wd, "uint4", "group", 128)
    np.testing.assert_array_equal(q.cpu().numpy(), eq)
    assert s.cpu().numpy().tobytes() == es.tobytes()
    np.testing.assert_array_equal(z.cpu().numpy(), ez)
    qp, s, z = ops.rtn_quantize(wd, "uint4", "group", 128, layout="kn_packed4")
    np.testing.assert_array_equal(qp.cpu().numpy().reshape(-1), O.pack_nibbles(eq))
    assert s.cpu().numpy().tobytes() == es.tobytes()
    np.testing.assert_array_equal(z.cpu().numpy(), ez)
    q8, s8, z8 = ops.rtn_quantize(wd, "int8", "group", 64, True)          # g = 64: sixteen lanes per row piece, two groups per 128 rows
    e8, es8, ez8 = O.rtn_quantize(w, "int8", "group", 64, True)
    np.testing.assert_array_equal(q8.cpu().numpy(), e8)
    assert s8.cpu().numpy().tobytes() == es8.tobytes()
    b8, _, _ = ops.rtn_quantize(wd, "uint8", "group", 64, layout="nbits")
    u8, us8, uz8 = O.rtn_quantize(w, "uint8", "group", 64)
    np.testing.assert_array_equal(b8.cpu().numpy(), O.matmul_nbits_layout(u8, us8, uz8, 64, 8)[0])


@pytest.mark.parametrize("qtype,g,sym,rr,clip", [("int4", 128, False, False, 1.0), ("int4", 64, True, False, 0.9),
                                                 ("int8", 128, True, True, 1.0), ("uint8", 32, True, False, 0.75),
                                                 ("uint4", 128, True, False, 1.0), ("int8", 32, False, True, 0.5),
                                                 ("int4", 512, False, False, 1.0), ("int8", 512, True, False, 0.8)])
def test_nbits_layout_signed_symmetric_clipped(ops, qtype, g, sym, rr, clip):
    """The blob kernels against the oracle for the grids MatMulNBits itself never sees (signed, symmetric, reduced,
    clipped): same integers as the [K, N] result, packed by qrules/_common.py:72-87's rule; heavy tails + zero groups."""
    rng = np.random.default_rng(23)
    k, n = 512, 296
    w = rng.standard_t(3, (k, n)).astype(np.float32)
    w[:g, :7] = 0.0                                    # all-zero groups: the tiny-scale guard (utils.py:262-263)
    eq, es, ez = O.rtn_quantize(w, qtype, "group", g, sym, rr, clip)
    bits = O.BITWIDTH[qtype]
    u = (eq.astype(np.int16) & (0xF if bits == 4 else 0xFF)).astype(np.uint8)   # two's-complement nibbles / bytes
    eb, _, _ = O.matmul_nbits_layout(u, es, ez, g, bits)
    b, s, z = ops.rtn_quantize(dev(w), qtype, "group", g, sym, rr, clip, layout="nbits")
    np.testing.assert_array_equal(b.cpu().numpy(), eb)
    assert s.cpu().numpy().tobytes() == es.tobytes()
    np.testing.assert_array_equal(z.cpu().numpy().reshape(-1), ez.reshape(-1))


def test_roundtrip_property_full_size(ops):
    """Size-independent property at BASELINE size: |dequant(q) - w| <= scale/2 (+1 ulp slack)
    wherever the value was not clipped, and every group hits both ends of the integer range."""
    import torch
    k, n, g = 4096, 11008, 128
    gen = torch.Generator(device="cuda").manual_seed(7)
    w = torch.randn((k, n), generator=gen, device="cuda", dtype=torch.float32)
    q, s, z = ops.rtn_quantize(w, "uint4", "group", g)
    dq = ops.dequantize(q, s, z, "uint4", mode="group", group=g)
    s_full = s.reshape(n, k // g).t().repeat_interleave(g, dim=0)      # [K, N]
    err = (dq - w).abs()
    assert bool((err <= s_full * 0.5000001 + 1e-7).all())
    qg = q.t().reshape(n, k // g, g)
    assert int(qg.amin(dim=2).max()) == 0 and int(qg.amax(dim=2).min()) == 15
    # idempotence: quantizing the dequantized weights reproduces the integers
    q2, _, _ = ops.rtn_quantize(dq, "uint4", "group", g)
    mism = int((q2 != q).sum())
    assert mism <= k * n // 1000          # ties at .5 may move by one after a round trip


def test_errors_are_loud(ops):
    import torch
    from onnx_quantize_amd.hip import OqHipError
    w = torch.zeros((64, 64), device="cuda")
    with pytest.raises(TypeError):
        ops.rtn_quantize(w.cpu(), "int8", "tensor")
    with pytest.raises(OqHipError, match="clip_ratio must be in"):
        ops.rtn_quantize(w, "int8", "tensor", clip_ratio=1.5)
    with pytest.raises(ValueError, match="cannot reshape"):
        ops.rtn_quantize(torch.zeros((10, 3), device="cuda"), "int8", "group", 4)


def test_back_to_back_launches_do_not_leak_between_calls(ops):
    """300 launches of different matrices back to back on one stream, no synchronisation in between: the
    staged [K/g, N] parameters + transpose launch (and the workspace reuse of the caching allocator) must never
    hand a later call the parameters of an earlier one.  Independent formula in torch (true division)."""
    import torch
    gen = torch.Generator(device="cuda").manual_seed(123)
    shapes = [(1024, 2304, 128), (2048, 1024, 64), (512, 4352, 128), (4096, 768, 32)]
    mats = [(torch.randn((k, n), generator=gen, device="cuda") * (0.5 + i), g) for i, (k, n, g) in enumerate(shapes * 3)]
    outs = []
    for rep in range(25):
        for w, g in mats:
            outs.append((w, g) + ops.rtn_quantize(w, "uint4", "group", g))
    torch.cuda.synchronize()
    for w, g, q, s, z in outs[::7] + outs[-12:]:
        k, n = w.shape
        wg = w.t().reshape(n, k // g, g)
        lo = wg.amin(dim=2).clamp(max=0)
        hi = wg.amax(dim=2).clamp(min=0)
        es = torch.div(hi - lo, torch.full_like(hi, 15.0))      # tensor / tensor: IEEE division (x / 15.0 would multiply by 1/15)
        ez = torch.clamp(0.0 - torch.div(lo, es), 0, 15).round()  # torch.round is half-to-even
        assert torch.equal(s.reshape(n, k // g), es)
        assert torch.equal(z.reshape(n, k // g).to(torch.float32), ez)


@pytest.mark.parametrize("layout", ["kn", "nbits"])
@pytest.mark.parametrize("qtype,g,sym", [("uint4", 128, False), ("int8", 64, True), ("uint8", 32, False)])
def test_batched_equals_per_matrix(ops, layout, qtype, g, sym):
    """One launch over a stack of matrices == the matrices quantized one by one (bit for bit)."""
    import torch
    gen = torch.Generator(device="cuda").manual_seed(77)
    w = torch.randn((5, 512, 768), generator=gen, device="cuda") * torch.tensor([0.5, 1, 2, 4, 8], device="cuda")[:, None, None]
    bq, bs, bz = ops.rtn_quantize_batched(w, qtype, g, sym, layout=layout)
    for b in range(w.shape[0]):
        q, s, z = ops.rtn_quantize(w[b], qtype, "group", g, sym, layout=layout)
        assert torch.equal(bq[b], q) and torch.equal(bs[b], s) and torch.equal(bz[b], z)


@pytest.mark.parametrize("k,n", [(1, 1), (1, 7), (2, 4), (7, 3), (33, 5), (128, 1), (128, 4), (130, 64), (256, 257), (512, 36), (96, 1028)])
def test_odd_shapes_all_strategies_against_oracle(ops, k, n):
    """Ragged and degenerate shapes (single row / column, N % 4 != 0, K not a multiple of the tile, group == K,
    group > K) through every strategy and both 4- and 8-bit grids: integers, zero points and scale bits equal the
    oracle's.  Group sizes are the divisors the reference's reshape accepts (utils.py:24)."""
    rng = np.random.default_rng(k * 1000 + n)
    w = (rng.standard_normal((k, n)) * rng.choice([1e-3, 1.0, 50.0])).astype(np.float32)
    configs = [("int8", "tensor", -1, True), ("uint8", "tensor", -1, False), ("int4", "channel", -1, False),
               ("uint4", "channel", -1, True), ("int8", "group", k, False), ("uint4", "group", 4 * k, False)]
    for g in (2, 16, 32, 64, 128):
        if k % g == 0:
            configs += [("uint4", "group", g, False), ("int8", "group", g, True)]
    for qtype, strategy, g, sym in configs:
        eq, es, ez = O.rtn_quantize(w, qtype, strategy, g, sym)
        q, s, z = ops.rtn_quantize(dev(w), qtype, strategy, g, sym)
        label = f"{k}x{n} {qtype} {strategy} g={g} sym={sym}"
        assert np.array_equal(q.cpu().numpy(), eq), label
        assert s.cpu().numpy().tobytes() == np.asarray(es, np.float32).tobytes(), label
        assert np.array_equal(z.cpu().numpy().reshape(-1), np.asarray(ez).reshape(-1)), label
        if strategy == "group" and k % min(g, k) == 0 and min(g, k) % 16 == 0 and (min(g, k) <= 256 or min(g, k) % 128 == 0) and n % 4 == 0:   # blob epilogues: fused (g <= 256) or two-pass (g % 128 == 0)
            b, s2, _ = ops.rtn_quantize(dev(w), qtype, "group", g, sym, layout="nbits")
            u = (eq.astype(np.int16) & (0xF if O.BITWIDTH[qtype] == 4 else 0xFF)).astype(np.uint8)
            eb, _, _ = O.matmul_nbits_layout(u, es, ez, min(g, k), O.BITWIDTH[qtype])
            assert np.array_equal(b.cpu().numpy(), eb), label + " (blob)"



@pytest.mark.gpu
@pytest.mark.parametrize("qtype,symmetric,reduce_range,clip", [("int8", True, False, 1.0), ("uint8", False, False, 1.0), ("int4", False, False, 0.9),
                                                                ("uint4", False, True, 1.0), ("int8", False, True, 0.75), ("int4", True, False, 1.0)])
def test_tensor_strategy_for_many_weights_in_one_call(qtype, symmetric, reduce_range, clip):
    """oq_rtn_tensor_many_f32: every tensor gets exactly what `_rtn_quantize(strategy=tensor)` gives it (rtn.py:54-109),
    whatever the mix of shapes, incl. a 1-element tensor, odd sizes (scalar head / tail paths) and an all-zero one."""
    import torch
    from onnx_quantize_amd.hip import ops
    rng = np.random.default_rng(11)
    shapes = [(640, 2048), (2048, 640), (640, 1024), (1, 1), (7, 13), (333, 129), (64, 64), (5, 4099)]
    hosts = [(rng.standard_normal(s) * rng.uniform(0.01, 3)).astype(np.float32) for s in shapes]
    hosts.append(np.zeros((16, 16), np.float32))
    hosts.append(np.abs(rng.standard_normal((31, 17))).astype(np.float32) + 1)          # strictly positive: zero joins the range
    res = ops.rtn_quantize_tensor_many([torch.from_numpy(h).cuda() for h in hosts], qtype, symmetric, reduce_range, clip)
    assert len(res) == len(hosts)
    for h, (q, s, z) in zip(hosts, res):
        eq, es, ez = O.rtn_quantize(h, qtype, "tensor", -1, symmetric, reduce_range, clip)
        assert q.shape == h.shape and s.shape == () and z.shape == ()
        np.testing.assert_array_equal(q.cpu().numpy(), eq)
        assert s.cpu().numpy().tobytes() == np.asarray(es, np.float32).tobytes() and int(z.cpu()) == int(ez)
    # and equals the one-tensor entry point bit for bit
    q1, s1, z1 = ops.rtn_quantize(torch.from_numpy(hosts[0]).cuda(), qtype, "tensor", -1, symmetric, reduce_range, clip)
    assert torch.equal(q1, res[0][0]) and torch.equal(s1.reshape(()), res[0][1]) and torch.equal(z1.reshape(()), res[0][2])


@pytest.mark.gpu
def test_tensor_many_errors_are_loud():
    import torch
    from onnx_quantize_amd.hip import ops
    assert ops.rtn_quantize_tensor_many([], "int8") == []
    with pytest.raises(NotImplementedError):
        ops.rtn_quantize_tensor_many([torch.zeros(4, device="cuda")], "int32")
    with pytest.raises(TypeError):
        ops.rtn_quantize_tensor_many([torch.zeros(4)], "int8")
    with pytest.raises(ValueError, match="zero-size"):
        ops.rtn_quantize_tensor_many([torch.zeros(0, device="cuda")], "int8")


@pytest.mark.gpu
@pytest.mark.parametrize("qtype", ["int4", "uint4", "int8", "uint8"])
@pytest.mark.parametrize("r,c,mode,group", [(64, 256, "tensor", 1), (33, 1028, "row", 1), (70, 260, "col", 1), (96, 132, "group", 32),
                                            (256, 1024, "group", 128), (40, 8, "group", 8), (129, 516, "group", 43), (1, 4096, "tensor", 1),
                                            (37, 63, "col", 1), (50, 1030, "row", 1)])
def test_elementwise_quantize_dequantize_every_parameter_mode(qtype, r, c, mode, group):
    """oq_quantize_f32 / oq_dequantize_f32 (utils.py:72-79, :130-132) with parameters per tensor / row / column / row
    group, on shapes that take the tiled fast path (C % 4 == 0) and on ones that take the per-element kernel, row
    counts that are no multiple of the 32-row tile, groups that straddle tiles: bit-equal to the oracle's arithmetic."""
    import torch
    from onnx_quantize_amd.hip import ops
    rng = np.random.default_rng(r * 1000 + c)
    x = (rng.standard_normal((r, c)) * 3).astype(np.float32)
    qmin, qmax = O.qrange(qtype, False, False)
    nparam = {"tensor": 1, "row": r, "col": c, "group": (r // group) * c}[mode]
    scale = rng.uniform(0.01, 0.3, size=nparam).astype(np.float32)
    zp = rng.integers(qmin, qmax + 1, size=nparam).astype(np.int32)
    if mode == "tensor":
        s_full, z_full = np.full((r, c), scale[0], np.float32), np.full((r, c), zp[0], np.int32)
    elif mode == "row":
        s_full, z_full = np.repeat(scale[:, None], c, 1), np.repeat(zp[:, None], c, 1)
    elif mode == "col":
        s_full, z_full = np.repeat(scale[None, :], r, 0), np.repeat(zp[None, :], r, 0)
    else:   # entry n * (R / g) + kg
        kg = np.arange(r) // group
        idx = np.arange(c)[None, :] * (r // group) + kg[:, None]
        s_full, z_full = scale[idx], zp[idx]
    eq = np.clip(np.rint(x / s_full).astype(np.int32) + z_full, qmin, qmax)
    edq = (eq.astype(np.float32) - z_full.astype(np.float32)) * s_full
    xd, sd, zd = torch.from_numpy(x).cuda(), torch.from_numpy(scale).cuda(), torch.from_numpy(zp).cuda()
    q = ops.quantize(xd, sd, zd, qtype, False, False, mode, group)
    np.testing.assert_array_equal(q.cpu().numpy().astype(np.int32), eq)
    dq = ops.dequantize(q, sd, zd, qtype, mode, group)
    assert dq.cpu().numpy().tobytes() == edq.astype(np.float32).tobytes()
    # a row-strided input (leading dimension > C) goes through the same kernels
    if c % 4 == 0:
        wide = torch.zeros((r, c + 8), device="cuda")
        wide[:, :c] = xd
        assert torch.equal(ops.quantize(wide[:, :c], sd, zd, qtype, False, False, mode, group), q)


@pytest.mark.gpu
@pytest.mark.parametrize("k,n,g", [(6, 4, 4), (20, 12, 16), (100, 8, 32), (96, 40, 64)])
def test_ragged_groups_roundtrip_against_the_oracle(k, n, g):
    """Groups that do not divide K straddle columns exactly like `W.T.reshape(-1, g)` (utils.py:24): element (k, n) belongs
    to group (n*K + k) // g.  `_dequantize_array(preprocess=True, GROUP)` / `ops.dequantize(mode="group")` /
    `ops.quantize(mode="group")` must address the parameters that way (K=6, N=4, g=4, element (2, 1) uses
    parameter 1, not 2, when addressed as (k // g) + n * (K // g))."""
    import torch
    from onnx_quantize_amd import QuantizationStrategy
    from onnx_quantize_amd.algorithms import functional as F
    from onnx_quantize_amd.hip import ops
    w = (np.random.default_rng(k * n + g).standard_normal((k, n)) * 2).astype(np.float32)
    eq, es, ez = O.rtn_quantize(w, "uint8", "group", g)
    rows = O.to_rows(eq, "group", g)
    edq = O.from_rows(O.dequantize(rows, es, ez), eq, "group")
    q, s, z = ops.rtn_quantize(torch.from_numpy(w).cuda(), "uint8", "group", g)
    np.testing.assert_array_equal(q.cpu().numpy(), eq)
    got = F._dequantize_array(eq, es, ez, preprocess=True, strategy=QuantizationStrategy.GROUP, group_size=g)
    assert got.tobytes() == np.ascontiguousarray(edq, np.float32).tobytes()
    dq = ops.dequantize(q, s, z, "uint8", mode="group", group=g)
    assert dq.cpu().numpy().tobytes() == np.ascontiguousarray(edq, np.float32).tobytes()
    rq = ops.quantize(torch.from_numpy(w).cuda(), s, z, "uint8", False, False, mode="group", group=g)
    np.testing.assert_array_equal(rq.cpu().numpy(), eq)
    assert (k * n) % 7 and k % 7
    with pytest.raises(ValueError, match="cannot reshape array"):
        ops.dequantize(q, s, z, "uint8", mode="group", group=7)


@pytest.mark.gpu
def test_dequantize_keeps_float_zero_points():
    """HQQ's zero points are floats (hqq.py:77-78) and utils.py:131 subtracts them as they are: 7.4 must not become 7."""
    import torch
    from onnx_quantize_amd import QuantizationStrategy
    from onnx_quantize_amd.algorithms import functional as F
    from onnx_quantize_amd.hip import ops
    rng = np.random.default_rng(9)
    k, n, g = 64, 24, 16
    q = rng.integers(0, 16, size=(k, n)).astype(np.uint8)
    s = rng.uniform(0.01, 0.2, size=(n * k // g, 1)).astype(np.float32)
    z = rng.uniform(0, 15, size=(n * k // g, 1)).astype(np.float32)
    exp = O.from_rows((O.to_rows(q, "group", g).astype(np.float32) - z) * s, q, "group")
    got = F._dequantize_array(q, s, z, preprocess=True, strategy=QuantizationStrategy.GROUP, group_size=g)
    assert got.tobytes() == np.ascontiguousarray(exp, np.float32).tobytes()
    dq = ops.dequantize(torch.from_numpy(q).cuda(), torch.from_numpy(s).cuda(), torch.from_numpy(z).cuda(), "uint4", mode="group", group=g)
    assert dq.cpu().numpy().tobytes() == np.ascontiguousarray(exp, np.float32).tobytes()
    trunc = O.from_rows((O.to_rows(q, "group", g).astype(np.float32) - np.trunc(z)) * s, q, "group")
    assert not np.array_equal(trunc, exp)


@pytest.mark.gpu
@pytest.mark.parametrize("strategy,g,layout", [("group", 128, "kn"), ("group", 128, "nbits"), ("group", 32, "kn"), ("channel", -1, "kn"),
                                               ("tensor", -1, "kn")])
def test_nan_weights_poison_their_range_like_numpy(strategy, g, layout):
    """np.min / np.max propagate NaN (utils.py:60-61): a NaN weight gives its group (channel, tensor) a NaN scale in the
    reference.  The kernels' reductions are NaN-propagating too (v_minimum3_f32 / v_maximum3_f32), so the corruption shows
    up in the same scales; every group without a NaN stays bit-exact.  (What integer a NaN itself becomes is
    implementation-defined in NumPy and outside the parity claim.)"""
    import torch
    from onnx_quantize_amd.hip import ops
    w = np.random.default_rng(77).standard_normal((256, 96)).astype(np.float32)
    w[130, 5] = np.nan
    w[3, 40] = np.nan
    with np.errstate(all="ignore"):
        import warnings
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            eq, es, ez = O.rtn_quantize(w, "uint4", strategy, g)
    q, s, z = ops.rtn_quantize(torch.from_numpy(w).cuda(), "uint4", strategy, g, layout=layout)
    s, z = s.cpu().numpy(), z.cpu().numpy()
    bad = np.isnan(es)
    assert bad.any() and np.array_equal(np.isnan(s), bad)
    assert s[~bad].tobytes() == es[~bad].tobytes() and np.array_equal(z[~bad], ez[~bad])
    if layout == "kn" and strategy == "group":
        rows_ok = ~bad.reshape(-1)
        np.testing.assert_array_equal(O.to_rows(q.cpu().numpy(), "group", g)[rows_ok], O.to_rows(eq, "group", g)[rows_ok])


@pytest.mark.gpu
@pytest.mark.parametrize("k,n", [(4100, 6400), (8200, 3100), (4096, 11008), (130, 200004)])
def test_per_tensor_kernel_with_dropped_tiles_and_ragged_edges_vs_oracle(k, n):
    """More than three tiles per CU (tiles are dropped in phase A and read again in phase B), a last row tile whose second
    half lies entirely past K, a last column tile that is only partly filled, rows of phase B straddling half tiles; twice on
    the same state buffer."""
    import torch
    from onnx_quantize_amd.hip import ops
    rng = np.random.default_rng(k * 7 + n)
    w = (rng.standard_normal((k, n)) * 0.03).astype(np.float32)
    w[k - 1, n - 1] = 0.9                       # the maximum sits in the last, ragged tile
    w[0, 0] = -0.7
    wd = torch.from_numpy(w).cuda()
    for qtype, sym in (("int8", False), ("uint8", True)):
        for _ in range(2):
            q, s, z = ops.rtn_quantize(wd, qtype, "tensor", -1, sym)
        eq, es, ez = O.rtn_quantize(w, qtype, "tensor", -1, sym)
        np.testing.assert_array_equal(q.cpu().numpy(), eq)
        assert np.asarray(s.cpu().numpy()).tobytes() == np.asarray(es).tobytes() and int(z.cpu().numpy()) == int(ez)
    state = ops._rtn_state(1, wd.device)
    torch.cuda.synchronize()
    assert int(state.count_nonzero()) == 0


@pytest.mark.gpu
@pytest.mark.parametrize("k,n,strategy", [(8192, 260, "channel"), (4096, 2048, "tensor"), (6144, 512, "tensor")])
def test_nan_weights_in_the_ticketed_kernels_for_large_ranges(k, n, strategy):
    """The same for the streamed channel kernel (ranges taller than 4096 rows) and for the per-tensor kernel with tiles kept
    in registers / LDS and tiles read twice: the NaN travels through the key atomics (top key) or the arrival slots."""
    import torch
    from onnx_quantize_amd.hip import ops
    w = (np.random.default_rng(k + n).standard_normal((k, n)) * 0.1).astype(np.float32)
    w[k - 7, 5] = np.nan
    with np.errstate(all="ignore"):
        import warnings
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            eq, es, ez = O.rtn_quantize(w, "int8", strategy, -1)
    q, s, z = ops.rtn_quantize(torch.from_numpy(w).cuda(), "int8", strategy, -1)
    s, z = np.asarray(s.cpu().numpy()).reshape(-1), np.asarray(z.cpu().numpy()).reshape(-1)
    es, ez = np.asarray(es).reshape(-1), np.asarray(ez).reshape(-1)
    bad = np.isnan(es)
    assert bad.any() and np.array_equal(np.isnan(s), bad)
    assert s[~bad].tobytes() == es[~bad].tobytes() and np.array_equal(z[~bad], ez[~bad])
    if strategy == "channel":
        np.testing.assert_array_equal(q.cpu().numpy()[:, ~bad], eq[:, ~bad])


@pytest.mark.gpu
def test_more_than_2_31_elements():
    """Indexing is 64-bit wherever a flat offset can pass 2^31: a 32768 x 69632 weight (2.28e9 elements, 9.1 GB) through
    the blob and the [K, N] kernels; sampled column strips (incl. the last one) against the oracle."""
    import torch
    from onnx_quantize_amd.hip import ops
    k, n, g = 32768, 69632, 128
    assert k * n > 2**31
    gen = torch.Generator(device="cuda").manual_seed(1)
    w = torch.empty((k, n), device="cuda")
    for r0 in range(0, k, 4096):                                    # generated in slabs: no 9 GB temporaries
        w[r0:r0 + 4096].normal_(generator=gen)
    blob, s, z = ops.rtn_quantize(w, "uint4", "group", g, layout="nbits")
    q, s2, z2 = ops.rtn_quantize(w, "uint4", "group", g)
    assert torch.equal(s, s2) and torch.equal(z, z2)
    kg = k // g
    for c0 in (0, 32768, 65536 - 32, n - 32):
        strip = w[:, c0:c0 + 32].cpu().numpy()
        eq, es, ez = O.rtn_quantize(strip, "uint4", "group", g)
        np.testing.assert_array_equal(q[:, c0:c0 + 32].cpu().numpy(), eq)
        got_s = s.reshape(n, kg)[c0:c0 + 32].cpu().numpy().reshape(-1, 1)
        got_z = z.reshape(n, kg)[c0:c0 + 32].cpu().numpy().reshape(-1, 1)
        assert got_s.tobytes() == es.tobytes() and np.array_equal(got_z, ez)
        eb, _, _ = O.matmul_nbits_layout(eq, es, ez, g, 4)
        np.testing.assert_array_equal(blob[c0:c0 + 32].cpu().numpy(), eb)
    del w, blob, q
    torch.cuda.empty_cache()


@pytest.mark.gpu
@pytest.mark.parametrize("strategy,k,n", [("channel", 8192, 1280), ("tensor", 4096, 8448), ("channel", 4224, 11008)])
def test_tie_bands_in_the_streamed_and_the_parked_paths(strategy, k, n):
    """The exact-division redo of a row GROUP (round 6: one decision per two / four rows) in the kernels that small shapes never
    reach: `rtn_resident_stream` (ranges taller than 4096 rows, or >= 1024 tiles) and `rtn_tensor_onepass` with more tiles than
    the chip keeps (1056 > 1024: phase B and all three parks).  Weights are multiples of a quarter of the step the range gives
    -- every other one sits exactly on a rounding tie -- plus a sprinkle of ordinary values; the whole result against the oracle."""
    import torch
    from onnx_quantize_amd.hip import ops
    r = np.random.default_rng(k + n)
    w = (r.integers(-510, 511, size=(k, n)) * 0.25).astype(np.float32)         # int8 range 255 levels over [-127.5, 127.5]: scale 1, ties at .5
    w[0, :] = -127.5
    w[1, :] = 127.5
    mask = r.random((k, n)) < 0.05
    w[mask] = np.clip(r.standard_normal(int(mask.sum())).astype(np.float32) * 40, -127.5, 127.5)
    w[0, :] = -127.5
    w[1, :] = 127.5
    eq, es, ez = O.rtn_quantize(w, "int8", strategy)
    q, s, z = ops.rtn_quantize(torch.from_numpy(w).cuda(), "int8", strategy)
    assert s.cpu().numpy().reshape(np.shape(es)).tobytes() == np.asarray(es, np.float32).tobytes()
    np.testing.assert_array_equal(z.cpu().numpy().reshape(np.shape(ez)), ez)
    np.testing.assert_array_equal(q.cpu().numpy(), eq)
    t = w / np.float32(np.asarray(es).reshape(-1)[0])
    ties = np.abs(np.abs(t - np.floor(t)) - 0.5) < 1e-3
    assert ties.mean() > 0.2                                                    # the data does what the docstring says


@pytest.mark.gpu
def test_per_tensor_at_the_tile_limit_of_the_one_pass_kernel():
    """`rtn_tensor_onepass` names a kept tile as tile + 1 in 15 bits of its arrival slot: 32 766 tiles (128 x 256 each) are the
    most it takes, a tensor of more goes to the three-launch path.  One tensor exactly at the limit (129 x 254 tiles = 1.07e9
    elements) and one a tile row above it: scale / zero point against the oracle's qparams of the tensor's range, sampled row
    strips (first, one in the middle of a tile, the last rows) against the oracle's K1 with those parameters, and the per-row
    sums of ALL integers against a plain torch evaluation of the same formula (IEEE division)."""
    import torch
    from onnx_quantize_amd.hip import ops
    n = 254 * 256
    for k in (129 * 128, 130 * 128 - 3):
        gen = torch.Generator(device="cuda").manual_seed(k)
        w = torch.empty((k, n), device="cuda")
        for r0 in range(0, k, 2048):                                    # generated in slabs: no 4 GB temporaries
            w[r0:r0 + 2048].normal_(generator=gen)
        w[k // 2, 5] = 9.5                                              # the range sits in the middle and at the very end
        w[k - 1, n - 1] = -8.25
        q, s, z = ops.rtn_quantize(w, "int8", "tensor")
        _, es, ez = O.rtn_quantize(np.array([[-8.25, 9.5, 0.0, 1.0]], dtype=np.float32), "int8", "tensor")      # a tensor of the same range
        es, ez = np.float32(np.asarray(es).reshape(-1)[0]), int(np.asarray(ez).reshape(-1)[0])
        assert np.float32(s.cpu().numpy().reshape(-1)[0]).tobytes() == es.tobytes() and int(z.cpu().numpy().reshape(-1)[0]) == ez
        for r0 in (0, 128 * 64 + 37, k - 9):
            strip = w[r0:r0 + 9].cpu().numpy()
            np.testing.assert_array_equal(q[r0:r0 + 9].cpu().numpy(), O.quantize(strip, es, ez, "int8", False, False))
        scale_t = torch.tensor(float(es), dtype=torch.float32, device="cuda")
        for r0 in range(0, k, 2048):                                    # every integer, slab by slab
            ref = torch.clamp(torch.round(w[r0:r0 + 2048] / scale_t) + ez, -128, 127).to(torch.int8)
            assert torch.equal(q[r0:r0 + 2048].view(torch.int8), ref), r0
        del w, q
        torch.cuda.empty_cache()


@pytest.mark.gpu
def test_kernel_emitted_wire_format_equals_the_reference_function_output():
    """tests/golden/nbits.*: what `_prepare_for_matmul_nbits` (qrules/_common.py:65-123) made of the reference's own RTN /
    HQQ results, against what the kernels write directly (layout="nbits") + the zero-point packing kernel."""
    import torch
    from conftest import load_json, load_npz, synth_weight
    from onnx_quantize_amd.hip import ops
    G, cases = load_npz("nbits.npz"), load_json("nbits.json")["cases"]
    for c in cases:
        key, g, k, n = c["key"], c["group_size"], c["k"], c["n"]
        w = torch.from_numpy(synth_weight("normal", c["seed"], k, n)).cuda()
        blocks = k // g
        if c["float_zero_points"]:
            blob, s, z, _ = ops.hqq_quantize(w, g, layout="nbits")
            # HQQ parity is tolerance-based: compare through the unpacked integers
            eb = G[key + "_blob"]
            lo, hi = (blob.cpu().numpy() & 15).astype(np.int32), (blob.cpu().numpy() >> 4).astype(np.int32)
            elo, ehi = (eb & 15).astype(np.int32), (eb >> 4).astype(np.int32)
            assert blob.shape == eb.shape and max(np.abs(lo - elo).max(), np.abs(hi - ehi).max()) <= 1
            assert (np.count_nonzero(lo != elo) + np.count_nonzero(hi != ehi)) <= 2e-3 * 2 * eb.size
            np.testing.assert_allclose(z.reshape(n, blocks).cpu().numpy(), G[key + "_zp"], atol=2e-5)
            assert s.reshape(n, blocks).cpu().numpy().tobytes() == G[key + "_scale"].tobytes()
            continue
        blob, s, z = ops.rtn_quantize(w, c["qtype"], "group", g, layout="nbits")
        np.testing.assert_array_equal(blob.cpu().numpy(), G[key + "_blob"])
        assert s.reshape(n, blocks).cpu().numpy().tobytes() == G[key + "_scale"].tobytes()
        if c["qtype"] == "uint4" and blocks > 1:
            np.testing.assert_array_equal(ops.pack_zero_points_u4(z, n, blocks).cpu().numpy(), G[key + "_zp"])
        else:
            np.testing.assert_array_equal(z.reshape(n, -1).cpu().numpy(), G[key + "_zp"])


@pytest.mark.gpu
def test_prepare_for_matmul_nbits_mirror_equals_the_reference_function():
    """`wire_format._prepare_for_matmul_nbits` (NumPy in / out through oq_pack_matmul_nbits + oq_pack_zero_points_u4) on
    the very inputs the reference function got (tests/golden/nbits.*): identical outputs, dtypes and shapes; and the
    stand-alone blob kernel on shapes with ragged 64 x 64 tiles."""
    import torch
    from conftest import load_json, load_npz
    from onnx_quantize_amd import HqqConfig, QConfig, QWeightArgs, QuantType
    from onnx_quantize_amd.hip import ops
    from onnx_quantize_amd.wire_format import _prepare_for_matmul_nbits
    G, cases = load_npz("nbits.npz"), load_json("nbits.json")["cases"]
    for c in cases:
        key = c["key"]
        extra = {"algorithm": HqqConfig()} if c["float_zero_points"] else {}
        qc = QConfig(weights=QWeightArgs(dtype=QuantType.from_string(c["qtype"]), group_size=c["group_size"], strategy="group", **extra))
        b, s, z = _prepare_for_matmul_nbits(G[key + "_q"], G[key + "_s"], G[key + "_z"], qc)
        for got, name in ((b, "_blob"), (s, "_scale"), (z, "_zp")):
            exp = G[key + name]
            assert got.dtype == exp.dtype and got.shape == exp.shape and got.tobytes() == exp.tobytes(), (key, name)
    rng = np.random.default_rng(0)
    for k, n, g, bits in ((96, 70, 16, 4), (130 * 2, 33, 26, 4), (192, 129, 64, 8), (80, 5, 80, 8), (2048, 300, 128, 4)):
        q = rng.integers(0, 2**bits, size=(k, n), dtype=np.uint8)
        eb, _, _ = O.matmul_nbits_layout(q, np.ones(n * k // g, np.float32), np.zeros((n * k // g, 1), np.uint8), g, bits)
        np.testing.assert_array_equal(ops.pack_matmul_nbits(torch.from_numpy(q).cuda(), g, bits).cpu().numpy(), eb)


@pytest.mark.gpu
def test_random_configurations_against_the_oracle():
    """Property test (hypothesis, fixed seed): random shapes, types, strategies, group sizes, symmetric / reduce_range /
    clip_ratio, value distributions with planted zeros, ties and huge / tiny magnitudes -- every output of
    `ops.rtn_quantize` (both layouts where the blob applies) equals the oracle's bit for bit."""
    import os
    import torch
    from hypothesis import HealthCheck, given, seed, settings, strategies as st
    from onnx_quantize_amd.hip import ops
    # a longer walk on request (a GPU box with minutes to spare): OQ_TEST_FUZZ_EXAMPLES=5000 OQ_TEST_FUZZ_WIDE=1 [OQ_TEST_FUZZ_SEED=n]
    fuzz_examples = int(os.environ.get("OQ_TEST_FUZZ_EXAMPLES", "300"))
    fuzz_wide = os.environ.get("OQ_TEST_FUZZ_WIDE", "0") == "1"

    @st.composite
    def case(draw):
        qtype = draw(st.sampled_from(["int4", "uint4", "int8", "uint8"]))
        strategy = draw(st.sampled_from(["tensor", "channel", "group"]))
        if strategy == "group":
            g = draw(st.sampled_from([2, 8, 16, 24, 32, 64, 96, 128, 200, 256]))
            k = g * draw(st.integers(1, 6))
        else:
            g, k = -1, draw(st.integers(1, 400))
        n = draw(st.integers(1, 300))
        if fuzz_wide and draw(st.integers(0, 3)) == 0:      # OQ_TEST_FUZZ_WIDE=1: widths where the block-order rules of rtn.hip switch
            n = draw(st.sampled_from([2048, 4096, 4100, 4224, 8192, 8196, 12288, 16384, 16388])) + draw(st.sampled_from([0, 0, 4, 128]))
            k = min(k, 2 * g) if strategy == "group" else min(k, 64)
        sym, red = draw(st.booleans()), draw(st.booleans())
        clip = draw(st.sampled_from([1.0, 0.9, 0.5, 0.999]))
        kind = draw(st.sampled_from(["normal", "wide", "tiny", "ties", "zeros", "positive"]))
        return qtype, strategy, g, k, n, sym, red, clip, kind, draw(st.integers(0, 2**31 - 1))

    def make(kind, k, n, rs):
        r = np.random.default_rng(rs)
        w = r.standard_normal((k, n)).astype(np.float32)
        if kind == "wide":
            w *= np.float32(10.0) ** r.integers(-6, 7, size=(1, n)).astype(np.float32)
        elif kind == "tiny":
            w *= np.float32(1e-30)
        elif kind == "ties":
            w = (r.integers(-40, 41, size=(k, n)) * 0.5).astype(np.float32)
        elif kind == "zeros":
            w[r.random((k, n)) < 0.7] = 0
            w[:, ::3] = 0
        elif kind == "positive":
            w = np.abs(w) + 1
        return w

    @seed(20240601 if fuzz_examples == 300 else int(os.environ.get("OQ_TEST_FUZZ_SEED", "1")))
    @settings(max_examples=fuzz_examples, deadline=None, suppress_health_check=list(HealthCheck))
    @given(case())
    def run(c):
        qtype, strategy, g, k, n, sym, red, clip, kind, rs = c
        w = make(kind, k, n, rs)
        eq, es, ez = O.rtn_quantize(w, qtype, strategy, g, sym, red, clip)
        wd = torch.from_numpy(w).cuda()
        q, s, z = ops.rtn_quantize(wd, qtype, strategy, g, sym, red, clip)
        np.testing.assert_array_equal(q.cpu().numpy(), eq, err_msg=str(c))
        assert s.cpu().numpy().reshape(np.shape(es)).tobytes() == np.asarray(es, np.float32).tobytes(), c
        np.testing.assert_array_equal(z.cpu().numpy().reshape(np.shape(ez)), ez, err_msg=str(c))
        if strategy == "group" and g % 16 == 0 and n % 4 == 0 and qtype in ("uint4", "uint8", "int4", "int8"):
            try:
                blob, s2, z2 = ops.rtn_quantize(wd, qtype, strategy, g, sym, red, clip, layout="nbits")
            except Exception as e:  # noqa: BLE001 -- shapes the blob path declines must say so, never answer wrongly
                assert "NBITS" in str(e) or "unsupported" in str(e).lower(), (c, e)
                return
            bits = 4 if "4" in qtype else 8
            eb, _, _ = O.matmul_nbits_layout(eq.view(np.uint8) & (15 if bits == 4 else 255), es, ez, g, bits)
            np.testing.assert_array_equal(blob.cpu().numpy(), eb, err_msg=str(c))
            assert torch.equal(s2.reshape(-1), s.reshape(-1)) and torch.equal(z2.reshape(-1), z.reshape(-1))

    run()


@pytest.mark.parametrize("qtype,g,layout", [("uint4", 128, "nbits"), ("uint4", 64, "nbits"), ("int4", 128, "kn"), ("int8", 32, "kn"),
                                            ("uint8", 128, "nbits"), ("uint4", 256, "kn")])
def test_list_of_weights_in_one_call_equals_the_per_matrix_calls(qtype, g, layout):
    """oq_rtn_quantize_ptrs_f32: a mixed-shape list -- repeated shapes, gemma-3-270m and Llama slices,
    a strided view -- through `ops.rtn_quantize_many` gives, per matrix, the bits of `ops.rtn_quantize`."""
    import torch
    from onnx_quantize_amd.hip import ops

    gen = torch.Generator(device="cuda").manual_seed(17)
    shapes = [(512, 384), (640, 2048), (512, 384), (2048, 640), (640, 1024), (512, 384), (1024, 4096), (640, 2048), (256, 132)]
    ws = [torch.randn(sh, generator=gen, device="cuda") * (0.01 + 0.1 * i) for i, sh in enumerate(shapes)]
    big = torch.randn((512, 1024), generator=gen, device="cuda")
    ws.append(big[:, :384])                                              # row-major view with ldw = 1024: its own shape group
    ws = [w for w in ws if w.shape[0] % g == 0]
    many = ops.rtn_quantize_many(ws, qtype, g, layout=layout)
    assert len(many) == len(ws)
    for w, (q, s, z) in zip(ws, many):
        q1, s1, z1 = ops.rtn_quantize(w, qtype, "group", g, layout=layout)
        assert q.shape == q1.shape and q.dtype == q1.dtype and torch.equal(q, q1)
        assert s.shape == s1.shape and torch.equal(s, s1)
        assert z.shape == z1.shape and z.dtype == z1.dtype and torch.equal(z, z1)
    sym = ops.rtn_quantize_many(ws[:3], qtype, g, symmetric=True, reduce_range=True, clip_ratio=0.9, layout=layout)
    for w, (q, s, z) in zip(ws[:3], sym):
        q1, s1, z1 = ops.rtn_quantize(w, qtype, "group", g, True, True, 0.9, layout=layout)
        assert torch.equal(q, q1) and torch.equal(s, s1) and torch.equal(z, z1)
    assert ops.rtn_quantize_many([], qtype, g) == []
    with pytest.raises(ValueError):
        ops.rtn_quantize_many([torch.zeros((g + g // 2, 8), device="cuda")], qtype, g)     # K % g != 0


@pytest.mark.gpu
@pytest.mark.parametrize("layout", ["kn", "nbits"])
def test_ptrs_entry_point_with_outputs_anywhere_in_memory(layout):
    """oq_hip.h promises a list of matrices AND outputs 'anywhere in device memory'.  ADVICE r03: with the staged [K,N]
    parameter path the transpose launch wrote entry z of a launch to scale_out(first entry) + z * stride, which is only
    right for the stacked buffers `ops.rtn_quantize_many` happens to build.  Here the C entry point is called directly
    with separately allocated outputs in scrambled order (and guard bytes around every parameter array)."""
    import ctypes as C

    import numpy as np
    import torch
    from onnx_quantize_amd.hip import _lib as L, ops

    lib = L.load()
    k, n, g, count = 512, 384, 64, 5
    gen = torch.Generator(device="cuda").manual_seed(5)
    ws = [torch.randn((k, n), generator=gen, device="cuda") * (0.05 + 0.2 * i) for i in range(count)]
    per = n * k // g
    qshape = (k, n) if layout == "kn" else (n, k // g, g // 2)
    # outputs allocated one by one, in an order unrelated to the table's, with a guard row in front of and behind the payload
    order = [3, 0, 4, 2, 1]
    bufs = {}
    for i in order:
        bufs[i] = (torch.zeros(qshape, dtype=torch.uint8, device="cuda"),
                   torch.full((per + 64,), -7.0, dtype=torch.float32, device="cuda"),
                   torch.full((per + 64,), 0xAB, dtype=torch.uint8, device="cuda"))
        torch.empty((1000 + 37 * i,), device="cuda")            # shift the next allocation
    table = np.empty((count, 4), dtype=np.int64)
    for i in range(count):
        q, sc, zp = bufs[i]
        table[i] = (ws[i].data_ptr(), q.data_ptr(), sc.data_ptr() + 32 * 4, zp.data_ptr() + 32)
    table_dev = torch.from_numpy(table).cuda()
    nbytes = lib.oq_rtn_batched_workspace_bytes(count, k, n, g)
    wsb = torch.empty(nbytes, dtype=torch.uint8, device="cuda")
    lay = L.OQ_LAYOUT_KN if layout == "kn" else L.OQ_LAYOUT_NBITS
    L.check(lib.oq_rtn_quantize_ptrs_f32(C.c_void_p(table.ctypes.data), C.c_void_p(table_dev.data_ptr()), count, k, n, n,
                                         L.QTYPE_CODE["uint4"], g, 0, 0, 1.0, lay, C.c_void_p(wsb.data_ptr()), wsb.numel(),
                                         C.c_void_p(torch.cuda.current_stream().cuda_stream)))
    torch.cuda.synchronize()
    for i in range(count):
        q1, s1, z1 = ops.rtn_quantize(ws[i], "uint4", "group", g, layout=layout)
        q, sc, zp = bufs[i]
        assert torch.equal(q.reshape(-1), q1.reshape(-1).view(torch.uint8)), f"entry {i}: integers"
        assert torch.equal(sc[32:32 + per], s1.reshape(-1)) and torch.equal(zp[32:32 + per], z1.reshape(-1)), f"entry {i}: parameters"
        assert bool((sc[:32] == -7.0).all()) and bool((sc[32 + per:] == -7.0).all()), f"entry {i}: scale guard overwritten"
        assert bool((zp[:32] == 0xAB).all()) and bool((zp[32 + per:] == 0xAB).all()), f"entry {i}: zero-point guard overwritten"
    # an empty shape is an argument error, not a division by zero
    st = lib.oq_rtn_quantize_ptrs_f32(C.c_void_p(table.ctypes.data), C.c_void_p(table_dev.data_ptr()), count, k, 0, 0,
                                      L.QTYPE_CODE["uint4"], g, 0, 0, 1.0, lay, C.c_void_p(wsb.data_ptr()), wsb.numel(), C.c_void_p(0))
    assert st == L.OQ_ERR_INVALID_ARGUMENT
    with pytest.raises(ValueError):
        ops.rtn_quantize_many([torch.zeros((g, 0), device="cuda")], "uint4", g)


@pytest.mark.parametrize("qtype", ["int4", "uint4"])
@pytest.mark.parametrize("shape,g", [((256, 1024), 128), ((128, 264), 16), ((512, 132), 64), ((96, 520), 32), ((512, 2056), 256),
                                     ((384, 256), 128), ((64, 8), 2)])
def test_packed_int4_kn_layout_is_the_reference_pack_of_the_kn_result(ops, qtype, shape, g):
    """OQ_LAYOUT_KN_PACKED4: the fused group kernel's epilogue writes core/_pack.py:8-22's serialisation of
    the [K, N] result directly -- N % 8 == 0 takes the dword stores (lane pairs exchange their byte pairs), N % 8 == 4 the
    two-byte ones, RPW != 16 (g = 2, 32 with 4 waves, 256) the generic form.  Checked against the oracle's `pack_nibbles` of
    the oracle's integers (pinned by the reference's own packing KATs) and against the unpacked GPU result."""
    k, n = shape
    rng = np.random.default_rng(k * 7 + n + g)
    w = rng.standard_t(3, size=(k, n)).astype(np.float32)
    for sym, red in ((False, False), (True, False), (False, True)):
        qp, s, z = ops.rtn_quantize(dev(w), qtype, "group", g, sym, red, 1.0, layout="kn_packed4")
        eq, es, ez = O.rtn_quantize(w, qtype, "group", g, sym, red)
        assert qp.shape == (k, n // 2) and qp.dtype.is_floating_point is False
        np.testing.assert_array_equal(qp.cpu().numpy().reshape(-1), O.pack_nibbles(eq))
        assert s.cpu().numpy().tobytes() == es.tobytes()
        np.testing.assert_array_equal(z.cpu().numpy(), ez)
        q1, _, _ = ops.rtn_quantize(dev(w), qtype, "group", g, sym, red, 1.0)
        assert ops.pack_nibbles(q1).cpu().numpy().tobytes() == qp.cpu().numpy().tobytes()


def test_packed_int4_kn_layout_refuses_what_it_cannot_write(ops):
    import torch
    w = torch.randn((256, 128), device="cuda")
    with pytest.raises(Exception, match="KN_PACKED4|4-bit"):
        ops.rtn_quantize(w, "int8", "group", 128, layout="kn_packed4")
    with pytest.raises(Exception, match="KN_PACKED4"):
        ops.rtn_quantize(w, "uint4", "channel", -1, layout="kn_packed4")
    with pytest.raises(ValueError):
        ops.rtn_quantize(w, "uint4", "group", 128, layout="nope")


@pytest.mark.parametrize("key", ["headline_int4_g128_packed", "headline_uint4_g128_packed"])
def test_packed_int4_full_size_digest(ops, key):
    """The 4096 x 11008 matrix, int4 / uint4 g128: the kernel's packed [K, N/2] bytes against the digest of
    `core/_pack.py::pack(_rtn_quantize(...))` run by the reference itself (make_golden.py::gen_digests)."""
    d = DIGESTS[key]
    w = synth_weight(d["kind"], d["seed"], d["k"], d["n"])
    q, s, z = ops.rtn_quantize(dev(w), d["qtype"], "group", d["group_size"], layout="kn_packed4")
    assert q.numel() == d["packed_bytes"]
    assert sha16(q.cpu().numpy()) == d["packed_sha"]
    assert sha16(s.cpu().numpy()) == d["s_sha"] and sha16(z.cpu().numpy()) == d["z_sha"]
    many = ops.rtn_quantize_many([dev(w), dev(w)], d["qtype"], d["group_size"], layout="kn_packed4")
    assert all(sha16(m[0].cpu().numpy()) == d["packed_sha"] for m in many)


def test_stateful_entry_point_leaves_its_state_zero_and_equals_the_plain_call(ops):
    """oq_rtn_quantize_stateful_f32: the caller's zeroed state replaces the clear launch of the one-read channel / tensor kernels;
    every call must leave it all-zero again (the last workgroup cleans up), whatever the shape, and give the bytes of
    oq_rtn_quantize_f32 (which clears a workspace region itself).  Interleaved shapes and strategies on one state buffer."""
    import ctypes as C

    import torch
    from onnx_quantize_amd.hip import _lib as L

    lib = L.load()
    gen = torch.Generator(device="cuda").manual_seed(77)
    state = torch.zeros(4 << 20, dtype=torch.uint8, device="cuda")
    stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    for (k, n), strategy, g in (((512, 1028), "channel", -1), ((384, 260), "tensor", -1), ((4096, 2048), "tensor", -1), ((1024, 516), "group", 512),
                                ((2048, 4096), "channel", -1), ((130, 2052), "tensor", -1), ((4096, 4096), "channel", -1),
                                ((8192, 516), "channel", -1), ((4096, 9000), "channel", -1), ((4224, 6400), "tensor", -1)):   # streamed kernel; dropped tiles
        w = torch.randn((k, n), generator=gen, device="cuda") * 0.3
        scode = L.STRATEGY_CODE[strategy]
        need = lib.oq_rtn_state_bytes(k, n, scode, g)
        assert 0 < need <= state.numel()
        count = {"tensor": 1, "channel": n}.get(strategy, n * k // max(g, 1))
        outs = []
        for use_state in (True, False):
            q = torch.empty((k, n), dtype=torch.int8, device="cuda")
            sc = torch.empty(count, dtype=torch.float32, device="cuda")
            zp = torch.empty(count, dtype=torch.int8, device="cuda")
            ws = torch.empty(lib.oq_rtn_workspace_bytes(k, n, scode, g, 0), dtype=torch.uint8, device="cuda")
            st = lib.oq_rtn_quantize_stateful_f32(C.c_void_p(w.data_ptr()), k, n, n, L.QTYPE_CODE["int8"], scode, g, 0, 0, 1.0, 0,
                                                  C.c_void_p(q.data_ptr()), C.c_void_p(sc.data_ptr()), C.c_void_p(zp.data_ptr()), L.OQ_LAYOUT_KN,
                                                  C.c_void_p(ws.data_ptr()), ws.numel(), C.c_void_p(state.data_ptr() if use_state else 0),
                                                  state.numel() if use_state else 0, stream)
            assert st == 0, lib.oq_last_error()
            outs.append((q, sc, zp))
        torch.cuda.synchronize()
        assert int(state.count_nonzero()) == 0, (k, n, strategy)          # cleaned up: ready for the next call
        assert all(torch.equal(a, b) for a, b in zip(*outs)), (k, n, strategy)
        eq, es, ez = O.rtn_quantize(w.cpu().numpy(), "int8", strategy, g)
        np.testing.assert_array_equal(outs[0][0].cpu().numpy(), eq)
    assert lib.oq_rtn_state_bytes(4096, 4096, L.STRATEGY_CODE["group"], 64) == 0        # 64-row groups: the fused kernel keeps no state
    assert lib.oq_rtn_state_bytes(4096, 4096, L.STRATEGY_CODE["group"], 128) >= 32 * 4096 * 6   # 128-row groups: the staging of the in-launch transposition


@pytest.mark.gpu
def test_ticketed_kernels_short_soak():
    """scripts/soak_resident.py for five seconds: bursts of channel / tensor / tall-group calls over mixed shapes without host
    synchronisation in between, one state buffer; every result equals the first one for its input, the state ends zero."""
    import os
    import subprocess
    import sys
    from conftest import ROOT
    r = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "soak_resident.py"), "5"], capture_output=True, text=True, timeout=240)
    assert r.returncode == 0 and "soak ok" in r.stdout, (r.stdout[-800:], r.stderr[-1500:])


def test_ticketed_calls_from_threads_on_their_own_streams(ops):
    """Per-tensor / per-channel RTN wait for sibling workgroups of their own launch; two such launches that overlapped on one
    device could stop each other for good (include/oq_hip.h).  The library chains them through an event (rtn_resident.hip::
    TicketChain), so threads on their own streams are safe: three threads, 36 calls each over the three ticketed kernels, every
    result equal to the single-thread one."""
    import threading
    import torch
    rng = np.random.default_rng(99)
    cases = [((4096, 2048), "int8", "tensor"), ((4096, 1024), "uint8", "channel"), ((8192, 512), "int8", "channel"), ((2048, 4096), "uint8", "tensor"),
             ((4096, 8200), "int4", "group128")]      # round 6: [K,N] group call whose appended blocks wait for its main blocks (in-launch transposition)

    def call(w, qtype, strategy):
        return ops.rtn_quantize(w, qtype, "group", 128, layout="kn_packed4") if strategy == "group128" else ops.rtn_quantize(w, qtype, strategy, -1)

    ws = [dev(rng.standard_normal(shape, dtype=np.float32)) for shape, _, _ in cases]
    want = []
    for w, (_, qtype, strategy) in zip(ws, cases):
        q, s_, z = call(w, qtype, strategy)
        want.append((q.cpu().numpy().copy(), s_.cpu().numpy().copy(), z.cpu().numpy().copy()))
    torch.cuda.synchronize()
    errors = []

    def worker(tid):
        try:
            stream = torch.cuda.Stream()
            with torch.cuda.stream(stream):
                for it in range(9):
                    for j in range(len(cases)):
                        i = (j + tid) % len(cases)
                        q, s_, z = call(ws[i], cases[i][1], cases[i][2])
                        if it % 3 == 2:        # a host round trip every third lap only: the other laps keep the streams full
                            stream.synchronize()
                            if not (np.array_equal(q.cpu().numpy(), want[i][0]) and s_.cpu().numpy().tobytes() == want[i][1].tobytes()
                                    and np.array_equal(z.cpu().numpy(), want[i][2])):
                                errors.append((tid, it, i))
                stream.synchronize()
        except Exception as e:   # noqa: BLE001
            errors.append((tid, repr(e)))

    threads = [threading.Thread(target=worker, args=(t,)) for t in range(3)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=120)
    assert not any(t.is_alive() for t in threads), "a thread is still waiting for the GPU"
    assert not errors, errors


def test_ticketed_strategies_under_graph_capture_take_the_three_launch_path(ops):
    """a captured ticketed kernel would be replayed unordered against every other ticketed launch of the device
    (an overlap is a hung GPU).  A capturing stream therefore gets the three-launch path (no tickets): same bits, and the
    graph can be replayed at will next to eager ticketed calls on other streams."""
    import torch
    rng = np.random.default_rng(7)
    w = dev(rng.standard_normal((2048, 1024), dtype=np.float32))
    for qtype, strategy in (("int8", "channel"), ("uint8", "tensor")):
        q0, s0, z0 = ops.rtn_quantize(w, qtype, strategy, -1)                       # eager: the ticketed kernels
        out = (torch.empty_like(q0), torch.empty_like(s0).reshape(-1), torch.empty_like(z0).reshape(-1))
        ops.rtn_quantize(w, qtype, strategy, -1, out=out)                           # workspaces exist before the capture
        torch.cuda.synchronize()
        for o in out:
            o.zero_()
        graph = torch.cuda.CUDAGraph()
        side = torch.cuda.Stream()
        with torch.cuda.graph(graph):
            ops.rtn_quantize(w, qtype, strategy, -1, out=out)
        for _ in range(3):
            graph.replay()
            with torch.cuda.stream(side):                                           # an eager ticketed call next to the replay
                q1, s1, z1 = ops.rtn_quantize(w, qtype, strategy, -1)
        torch.cuda.synchronize()
        assert torch.equal(out[0], q0) and out[1].cpu().numpy().tobytes() == s0.cpu().numpy().tobytes() and torch.equal(out[2].reshape(z0.shape), z0)
        assert torch.equal(q1, q0) and torch.equal(z1, z0)


@pytest.mark.gpu
def test_kn_layouts_with_the_parameters_transposed_inside_the_launch():
    """Round 6: with the caller's zeroed state the 128-row group call in the [K,N] layouts (bytes and packed nibbles, up to 64
    k-groups) stages its parameters in the state (self-validating words) and appended blocks transpose them inside the launch
    (rtn.hip::transposer_block).  Same bytes as the plain entry point (staged + a transpose launch), the state is zero again after
    every call, ragged widths, two passes of the transposer and calls in a row on one state buffer included."""
    import ctypes as C

    import torch
    from onnx_quantize_amd.hip import _lib as L

    lib = L.load()
    gen = torch.Generator(device="cuda").manual_seed(606)
    stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    state = torch.zeros(8 << 20, dtype=torch.uint8, device="cuda")
    cases = [(4096, 11008, "int4", "packed"), (4096, 11000, "uint4", "packed"), (2048, 5376, "int4", "packed"), (4096, 2824, "int4", "packed"),
             (512, 6664, "uint4", "packed"), (8192, 5376, "int4", "packed"),      # 64 k-groups: two passes of the transposer
             (4096, 11008, "int8", "kn"), (4096, 5388, "uint4", "kn"), (8192, 5376, "uint8", "kn"), (1024, 11008, "int4", "kn")]
    for (k, n, qtype, lay) in cases:
        g = 128
        groups = n * (k // g)
        scode = L.STRATEGY_CODE["group"]
        layout = L.OQ_LAYOUT_KN_PACKED4 if lay == "packed" else L.OQ_LAYOUT_KN
        assert 0 < lib.oq_rtn_state_bytes(k, n, scode, g) <= state.numel()
        ws = torch.empty(lib.oq_rtn_workspace_bytes(k, n, scode, g, 0) + 256, dtype=torch.uint8, device="cuda")
        for rep in range(3):
            w = torch.randn((k, n), generator=gen, device="cuda") * (0.1 + rep)
            got = []
            for use_state in (False, True):
                q = torch.full((k * n // 2 if lay == "packed" else k * n,), 0x5a, dtype=torch.uint8, device="cuda")
                sc = torch.full((groups,), -1.0, dtype=torch.float32, device="cuda")
                zp = torch.full((groups,), 0x33, dtype=torch.uint8, device="cuda")
                st = lib.oq_rtn_quantize_stateful_f32(C.c_void_p(w.data_ptr()), k, n, n, L.QTYPE_CODE[qtype], scode, g, 0, 0, 1.0, 0,
                                                      C.c_void_p(q.data_ptr()), C.c_void_p(sc.data_ptr()), C.c_void_p(zp.data_ptr()), layout,
                                                      C.c_void_p(ws.data_ptr()), ws.numel(), C.c_void_p(state.data_ptr() if use_state else 0),
                                                      state.numel() if use_state else 0, stream)
                assert st == 0, lib.oq_last_error()
                got.append((q, sc, zp))
            torch.cuda.synchronize()
            assert int(state.count_nonzero()) == 0, (k, n, rep)
            assert all(torch.equal(a, b) for a, b in zip(*got)), (k, n, qtype, lay, rep)
            if rep == 0:      # and the values are the oracle's
                eq, es, ez = O.rtn_quantize(w.cpu().numpy(), qtype, "group", g)
                assert es.reshape(-1).tobytes() == got[1][1].cpu().numpy().tobytes()
                np.testing.assert_array_equal(got[1][2].cpu().numpy().view(ez.dtype).reshape(-1), ez.reshape(-1))
                if lay == "kn":
                    np.testing.assert_array_equal(got[1][0].cpu().numpy().view(eq.dtype).reshape(eq.shape), eq)


@pytest.mark.gpu
def test_in_launch_transposition_random_shapes_against_the_oracle(ops):
    """The shapes the property test above never reaches (its groups are at most six per column and its matrices too small for staged
    parameters): 128-row groups x {4 ... 64} k-groups x wide and ragged widths through `ops.rtn_quantize` (stateful entry point: the
    parameters are transposed inside the launch when the rule takes the shape, by a second launch otherwise) against the oracle --
    integers, scales and zero points bit for bit; symmetric / reduce_range / clip_ratio drawn as well."""
    import torch
    rng = np.random.default_rng(20251005)
    for _ in range(14):
        kgroups = int(rng.choice([4, 8, 12, 16, 28, 32, 36, 64]))
        n = int(rng.choice([2820, 4100, 5376, 6664, 8196, 11000, 11008, 2304]))
        qtype = str(rng.choice(["int4", "uint4", "int8", "uint8"]))
        layout = "kn_packed4" if (qtype in ("int4", "uint4") and n % 2 == 0 and rng.random() < 0.6) else "kn"
        sym, red, clip = bool(rng.random() < 0.3), bool(rng.random() < 0.2), float(rng.choice([1.0, 0.9]))
        k = 128 * kgroups
        w = rng.standard_normal((k, n), dtype=np.float32) * np.float32(10.0) ** rng.integers(-3, 4, size=(1, n)).astype(np.float32)
        w[:, ::7] = 0                                                      # whole zero groups: the tiny-scale guard (scale 1: never a zero word)
        eq, es, ez = O.rtn_quantize(w, qtype, "group", 128, sym, red, clip)
        wd = torch.from_numpy(w).cuda()
        for rep in range(2):                                               # twice on the same state
            q, s, z = ops.rtn_quantize(wd, qtype, "group", 128, sym, red, clip, layout=layout)
            case = (kgroups, n, qtype, layout, sym, red, clip, rep)
            assert s.cpu().numpy().reshape(np.shape(es)).tobytes() == np.asarray(es, np.float32).tobytes(), case
            np.testing.assert_array_equal(z.cpu().numpy().reshape(np.shape(ez)), ez, err_msg=str(case))
            if layout == "kn":
                np.testing.assert_array_equal(q.cpu().numpy(), eq, err_msg=str(case))
            else:
                b = q.cpu().numpy().reshape(k, n // 2)
                lo, hi = (b & 0x0F).astype(np.int8), (b >> 4).astype(np.int8)
                if qtype == "int4":
                    lo, hi = ((lo ^ 8) - 8).astype(np.int8), ((hi ^ 8) - 8).astype(np.int8)
                full = np.empty((k, n), np.int8)
                full[:, 0::2], full[:, 1::2] = lo, hi
                np.testing.assert_array_equal(full.astype(np.int32), eq.astype(np.int32), err_msg=str(case))
