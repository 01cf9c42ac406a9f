"""The oracle (oracle/oq_oracle.py) against the golden vectors produced by the reference
itself (tests/golden/make_golden.py) and against the reference tests' known answers.
CPU only.  Bit-exact for integers; fp32 scales are compared bit-for-bit too, because the
oracle goes through the very same NumPy ufuncs."""
import hashlib

import numpy as np
import pytest

import oq_oracle as O
from conftest import load_json, load_npz, synth_weight


def sha16(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()[:16]


KATS = load_json("scalar_kats.json")


@pytest.mark.parametrize("qtype,sym,red,expected", KATS["qrange"])
def test_qrange_table(qtype, sym, red, expected):
    assert list(O.qrange(qtype, sym, red)) == expected


@pytest.mark.parametrize("vals,qtype,sym,exp_scale,exp_zp", KATS["qparams"])
@pytest.mark.parametrize("mse", [False, True])
def test_qparams_known_answers(vals, qtype, sym, exp_scale, exp_zp, mse):
    # reference: test/core/algorithms/test_rtn.py:21-72 (float64 inputs, tensor strategy)
    s, z = O.qparams_from_rows(np.array(vals), qtype, "tensor", sym, False, 1.0, mse,
                               np.float32, O.container_dtype(qtype))
    assert s.size == 1 and s > 0 and s.dtype == np.float32
    np.testing.assert_allclose(s, np.float32(exp_scale), rtol=1e-5)
    assert z.dtype == O.container_dtype(qtype) and int(z) == exp_zp


@pytest.mark.parametrize("vals,qtype,expected", KATS["pack"])
def test_pack_known_answers(vals, qtype, expected):
    # reference: test/core/test_pack.py:11-27, :59-75
    a = np.array(vals, dtype=np.int8 if qtype == "int4" else np.uint8)
    packed = O.pack_nibbles(a)
    assert packed.dtype == np.uint8 and packed.tolist() == expected
    back = O.unpack_nibbles(packed, a.shape, signed=(qtype == "int4"))
    np.testing.assert_array_equal(back, a)


def test_ema_known_answer():
    # reference: test/core/calibration/test_minmax_calibrator.py:127-144
    e = KATS["ema"]
    c = O.MinMaxOracle(momentum=e["momentum"])
    for b in e["batches"]:
        c.collect("t", np.array(b))
    assert np.isclose(c.data["t"][0], e["min"]) and np.isclose(c.data["t"][1], e["max"])
    lo, hi = c.compute_range("t")
    np.testing.assert_almost_equal(lo, e["min"])
    np.testing.assert_almost_equal(hi, e["max"])
    with pytest.raises(KeyError, match="No calibration data collected for 'nope'"):
        c.compute_range("nope")


def test_nbits_odd_block_zero_points():
    # reference: test/qrules/test_common.py:7-31
    g, blocks, n = 16, 5, 4
    r = np.random.default_rng(0)
    q = r.integers(0, 16, size=(g * blocks, n), dtype=np.uint8)
    s = r.random(size=(n * blocks,)).astype(np.float32)
    z = r.integers(0, 16, size=(n * blocks, 1), dtype=np.uint8)
    b, s2, pz = O.matmul_nbits_layout(q, s, z, g, 4)
    assert b.shape == (n, blocks, g // 2) and s2.shape == (n, blocks)
    assert pz.shape == (n, (blocks + 1) // 2)
    nib = np.empty((n, pz.shape[1] * 2), np.uint8)
    nib[:, ::2] = pz & 0x0F
    nib[:, 1::2] = pz >> 4
    np.testing.assert_array_equal(nib[:, :blocks], z.reshape(n, blocks))
    assert np.all(nib[:, blocks:] == 0x8)
    # B bytes: low nibble = even k inside the group
    col, blk = 2, 3
    grp = q[blk * g:(blk + 1) * g, col]
    np.testing.assert_array_equal(b[col, blk], grp[0::2] | (grp[1::2] << 4))


RTN_CASES = load_json("rtn_small.json")
RTN = load_npz("rtn_small.npz")


@pytest.mark.parametrize("case", RTN_CASES, ids=[c["id"] for c in RTN_CASES])
def test_rtn_small_bit_exact(case):
    cid = case["id"]
    q, s, z = O.rtn_quantize(RTN[f"{cid}_w"], case["qtype"], case["strategy"],
                             case["group_size"], case["symmetric"], case["reduce_range"],
                             case["clip_ratio"], False)
    gq, gs, gz = RTN[f"{cid}_q"], RTN[f"{cid}_s"], RTN[f"{cid}_z"]
    assert q.dtype == gq.dtype and q.shape == gq.shape
    np.testing.assert_array_equal(q, gq)
    assert s.dtype == np.float32 and s.shape == gs.shape
    assert s.tobytes() == gs.tobytes()
    assert z.dtype == gz.dtype and z.shape == gz.shape
    np.testing.assert_array_equal(z, gz)


MSE_CASES = load_json("rtn_mse.json")
MSE = load_npz("rtn_mse.npz")


@pytest.mark.parametrize("case", MSE_CASES, ids=[c["id"] for c in MSE_CASES])
def test_rtn_mse_bit_exact(case):
    cid = case["id"]
    w = MSE[f"{cid}_w"]
    rows = O.to_rows(w, case["strategy"], case["group_size"])
    lo, hi = O.min_max_mse(rows, case["qtype"], case["strategy"], case["symmetric"], False)
    assert np.asarray(lo).tobytes() == MSE[f"{cid}_lo"].tobytes()
    assert np.asarray(hi).tobytes() == MSE[f"{cid}_hi"].tobytes()
    q, s, z = O.rtn_quantize(w, case["qtype"], case["strategy"], case["group_size"],
                             case["symmetric"], False, 1.0, True)
    np.testing.assert_array_equal(q, MSE[f"{cid}_q"])
    assert s.tobytes() == MSE[f"{cid}_s"].tobytes()
    np.testing.assert_array_equal(z, MSE[f"{cid}_z"])


def test_elementwise_kernels_bit_exact():
    K = load_npz("kernels.npz")
    x = K["x"]
    lo = np.minimum(x.min(axis=1, keepdims=True), 0)
    hi = np.maximum(x.max(axis=1, keepdims=True), 0)
    for qtype in ("int4", "uint4", "int8", "uint8"):
        for sym in (False, True):
            for red in (False, True):
                tag = f"{qtype}_{int(sym)}{int(red)}"
                s, z = O.qparams(lo, hi, qtype, sym, red)
                assert s.tobytes() == K[f"qp_{tag}_s"].tobytes(), tag
                np.testing.assert_array_equal(z, K[f"qp_{tag}_z"])
                q = O.quantize(x, s, z, qtype, sym, red)
                np.testing.assert_array_equal(q, K[f"qp_{tag}_q"])
                assert O.dequantize(q, s, z).tobytes() == K[f"qp_{tag}_dq"].tobytes()
    for strategy, g in (("tensor", -1), ("channel", -1), ("group", 16)):
        out = O.dequantize(K[f"dq_{strategy}_q"], K[f"dq_{strategy}_s"], K[f"dq_{strategy}_z"],
                           rows_of=strategy, group_size=g)
        assert np.ascontiguousarray(out).tobytes() == K[f"dq_{strategy}_out"].tobytes()
    qb, bs, zp = O.quantize_bias(K["bias"], K["bias_xscale"], K["bias_wscale"])
    np.testing.assert_array_equal(qb, K["bias_q"])
    assert qb.dtype == np.int32 and zp == 0
    assert bs.tobytes() == K["bias_scale"].tobytes()


def test_minmax_sequences_bit_exact():
    M = load_npz("minmax.npz")
    for seq in load_json("minmax.json"):
        sid = seq["id"]
        cal = O.MinMaxOracle(momentum=seq["momentum"])
        for b in range(seq["batches"]):
            cal.collect("t", M[f"{sid}_b{b}"])
            assert np.asarray(cal.data["t"][0]).tobytes() == M[f"{sid}_b{b}_min"].tobytes()
            assert np.asarray(cal.data["t"][1]).tobytes() == M[f"{sid}_b{b}_max"].tobytes()
        lo, hi = cal.compute_range("t")
        assert lo.tobytes() == M[f"{sid}_lo"].tobytes() and hi.tobytes() == M[f"{sid}_hi"].tobytes()
        for qtype, sym in (("int8", False), ("uint8", False), ("int8", True)):
            s, z = O.qparams(lo, hi, qtype, sym, False)
            assert np.asarray(s).tobytes() == M[f"{sid}_{qtype}_{int(sym)}_scale"].tobytes()
            np.testing.assert_array_equal(z, M[f"{sid}_{qtype}_{int(sym)}_zp"])


GPTQ_CASES = load_json("gptq.json")
GPTQ = load_npz("gptq.npz")


@pytest.mark.parametrize("case", GPTQ_CASES, ids=[c["id"] for c in GPTQ_CASES])
def test_gptq_parity_mode_bit_exact(case):
    cid, d = case["id"], case["data"]
    q, s, z = O.gptq_quantize(GPTQ[f"{d}_w"], GPTQ[f"{d}_x"], case["qtype"], case["strategy"],
                              case["group_size"], case["symmetric"], case["reduce_range"],
                              case["clip_ratio"], case["block_size"], case["percdamp"],
                              case["actorder"], case["mse"], mode="parity")
    np.testing.assert_array_equal(q, GPTQ[f"{cid}_q"])
    assert q.dtype == GPTQ[f"{cid}_q"].dtype
    assert s.shape == GPTQ[f"{cid}_s"].shape and s.tobytes() == GPTQ[f"{cid}_s"].tobytes()
    np.testing.assert_array_equal(z, GPTQ[f"{cid}_z"])


def test_hessian_accumulate():
    x = GPTQ["b_x"]
    k = x.shape[-1]
    h, n = O.accumulate_hessian(x, np.zeros((k, k), np.float32), 0)
    assert n == int(GPTQ["b_nsamples"]) == x.shape[0]
    np.testing.assert_array_equal(h, GPTQ["b_h"])        # same BLAS, same order here
    h2, n2 = O.accumulate_hessian(x[:4], np.zeros((k, k), np.float32), 0)
    h2, n2 = O.accumulate_hessian(x[4:], h2, n2)
    np.testing.assert_allclose(h2, GPTQ["b_h_two_step"], rtol=0, atol=0)
    np.testing.assert_allclose(h2, h, rtol=0, atol=1e-6 * float(np.abs(h).max()))


def test_gptq_factor_properties():
    h = GPTQ["b_h"].copy()
    dead = np.diag(h) == 0
    assert dead.sum() == 3
    h[dead, dead] = 1
    u, ok = O.gptq_factor(h, 0.01)
    assert ok
    assert np.all(np.tril(u, -1) == 0)                   # upper factor: finding #1 of the survey
    hd = h.astype(np.float64)
    hd[np.diag_indices_from(hd)] += 0.01 * np.mean(np.diag(h))
    np.testing.assert_allclose(u.astype(np.float64).T @ u.astype(np.float64),
                               np.linalg.inv(hd), rtol=0, atol=5e-4 * np.abs(np.linalg.inv(hd)).max())
    u2, ok2 = O.gptq_factor(-np.eye(4, dtype=np.float32), 0.01)
    assert not ok2 and np.array_equal(u2, np.eye(4, dtype=np.float32))


def test_gptq_corrected_mode_differs_and_lowers_error():
    """The opt-in mode is real GPTQ: it must reduce the layer output error vs RTN."""
    w, x = GPTQ["b_w"], GPTQ["b_x"]
    args = ("int4", "group", 128, False, False, 1.0, 128, 0.01, False, False)
    qp, sp, zp = O.gptq_quantize(w, x, *args, mode="parity")
    qc, sc, zc = O.gptq_quantize(w, x, *args, mode="corrected")
    assert np.any(qp != qc)
    x2 = x.reshape(-1, x.shape[-1]).astype(np.float64)

    def out_err(q, s, z):
        dq = O.dequantize(q, s, z, rows_of="group", group_size=128).astype(np.float64)
        return np.linalg.norm(x2 @ dq - x2 @ w.astype(np.float64))
    assert out_err(qc, sc, zc) < out_err(qp, sp, zp)


DIGESTS = load_json("digests.json")


def test_config1_digest():
    d = DIGESTS["config1"]
    w = synth_weight(d["kind"], d["seed"], d["k"], d["n"])
    assert sha16(w) == d["w_sha"]
    q, s, z = O.rtn_quantize(w, d["qtype"], d["strategy"], d["group_size"], d["symmetric"])
    assert sha16(q) == d["q_sha"] and s.tobytes().hex() == d["scale_hex"] and int(z) == d["zp"]


@pytest.mark.parametrize("key", ["config2_asym", "config2_sym"])
def test_config2_digest(key):
    d = DIGESTS[key]
    w = synth_weight(d["kind"], d["seed"], d["k"], d["n"])
    assert sha16(w) == d["w_sha"]
    q, s, z = O.rtn_quantize(w, d["qtype"], d["strategy"], d["group_size"], d["symmetric"])
    assert (sha16(q), sha16(s), sha16(z)) == (d["q_sha"], d["s_sha"], d["z_sha"])
    assert s[0].tobytes().hex() == d["scale0_hex"]
    assert z[:4, 0].tolist() == d["zp_head"] and q[:4, 0].tolist() == d["q_head"]
    assert s.shape == (d["n"] * d["k"] // 128, 1)


def test_matmul_nbits_layout_vs_the_reference_function():
    """tests/golden/nbits.*: `_prepare_for_matmul_nbits` (qrules/_common.py:65-123) itself on RTN / HQQ results -- B blob,
    [N, K/g] scales, packed (or float, for HQQ) zero points -- reproduced byte for byte."""
    G, cases = load_npz("nbits.npz"), load_json("nbits.json")["cases"]
    assert len(cases) == 8
    for c in cases:
        key = c["key"]
        bits = 4 if c["qtype"] == "uint4" else 8
        b, s, z = O.matmul_nbits_layout(G[key + "_q"], G[key + "_s"], G[key + "_z"], c["group_size"], bits,
                                        zp_is_float=c["float_zero_points"])
        assert b.dtype == G[key + "_blob"].dtype and b.shape == G[key + "_blob"].shape
        np.testing.assert_array_equal(b, G[key + "_blob"])
        assert s.shape == G[key + "_scale"].shape and s.tobytes() == G[key + "_scale"].tobytes()
        assert z.dtype == G[key + "_zp"].dtype and z.shape == G[key + "_zp"].shape
        np.testing.assert_array_equal(z, G[key + "_zp"])
