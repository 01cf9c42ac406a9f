"""the on-device calibration driver against the reference's hold-everything flow
(core/_calibration/calibrate.py:204-385), restated in oracle/oq_oracle.py (`calibrate_flow`, `gptq_inputs`)."""
import numpy as np
import pytest

from oracle import oq_oracle as O
from onnx_quantize_amd import calibration_driver as D
from onnx_quantize_amd.config import QActivationArgs
from onnx_quantize_amd.dtypes import QuantType


# ----------------------------------------------------------------------------- host logic (no GPU)
@pytest.mark.parametrize("total,batch,num,shape", [(10, 2, 10, (5, 2)), (10, 5, 10, (2, 5)), (10, 10, 10, (1, 10)),
                                                   (10, 20, 10, (1, 10)), (10, 3, 10, (3, 3)), (10, 4, 100, (2, 4)),
                                                   (12, 4, 7, (1, 4)), (3, 1, 2, (2, 1))])
def test_prepare_calibration_data_batches_like_the_reference(total, batch, num, shape):
    """test_calibrate.py:139-176: the (batch_size, num_samples) grid, incl. 'drops excess samples' (10 // 3 = 3)."""
    data = np.arange(total * 6, dtype=np.float32).reshape(total, 2, 3)
    got = D.prepare_calibration_data(data, batch, num)
    exp = O.prepare_calibration_data(data, batch, num)
    assert got.shape == shape + (2, 3)
    np.testing.assert_array_equal(got, exp)
    assert np.shares_memory(got, data)                                  # a view, never a copy
    import torch
    t = D.prepare_calibration_data(torch.from_numpy(data), batch, num)
    np.testing.assert_array_equal(t.numpy(), exp)


def test_oracle_walk_order_equals_the_reference_calibrate_model():
    """tests/golden/calibrate.*: the reference's own `calibrate_model` (calibrate.py:310-385) run on prepared activation
    lists (make_golden.py::gen_calibrate replaces only the onnxruntime session).  The oracle's `calibrate_flow` gives the
    same (scale, zero point) for every node and kind -- including the cases where input AND output kinds are calibrated
    with momentum > 0, i.e. the output ranges are an EMA over the batch sequence seen twice -- bit for bit; `gptq_inputs`
    and `prepare_calibration_data` likewise."""
    from conftest import load_json, load_npz
    G, meta = load_npz("calibrate.npz"), load_json("calibrate.json")
    assert len(meta["cases"]) == 7
    chain_in, chain_out = ["X", "h1", "h2"], ["h1", "h2", "Y"]
    for c in meta["cases"]:
        key = c["key"]
        acts = [{n: G[f"{key}_b{b}_{n}"] for n in c["names"]} for b in range(c["batches"])]
        in_names = chain_in if c["kinds"] != "output" else []
        out_names = chain_out if c["kinds"] != "input" else []
        res = O.calibrate_flow(acts, in_names, out_names, c["momentum"], ("uint8", False, False) if in_names else None,
                               ("int8", True, False) if out_names else None)
        assert {k: sorted(v) for k, v in c["set"].items()} == {kind: sorted(n for (kd, n) in res if kd == kind) for kind in c["set"]}
        for (kind, name), (scale, zp) in res.items():
            assert np.asarray(scale).tobytes() == G[f"{key}_{kind}_{name}_scale"].tobytes(), (key, kind, name)
            assert int(zp) == int(G[f"{key}_{kind}_{name}_zp"]) and np.asarray(zp).dtype == G[f"{key}_{kind}_{name}_zp"].dtype
    acts = [{n: G[f"gptq_b{b}_{n}"] for n in chain_in} for b in range(meta["gptq_batches"])]
    whole = O.gptq_inputs(acts)
    for n in chain_in:
        np.testing.assert_array_equal(whole[n], G[f"gptq_input_{n}"])
    data = np.arange(10 * 3, dtype=np.float32).reshape(10, 3)
    for bs, ns in meta["prepare"]:
        np.testing.assert_array_equal(O.prepare_calibration_data(data, bs, ns), G[f"prep_{bs}_{ns}"])
        np.testing.assert_array_equal(D.prepare_calibration_data(data, bs, ns), G[f"prep_{bs}_{ns}"])


def test_random_calibration_data_matches_the_reference_recipe():
    """calibrate.py:127-147: default_rng(0), symbolic dims -> 1, ints in [0, 100), one generator for all inputs."""
    one = D.generate_random_calibration_data(12, [("X", ("N", 32), np.float32)])
    np.testing.assert_array_equal(one, np.random.default_rng(0).standard_normal((12, 32)).astype(np.float32))
    ids = D.generate_random_calibration_data(4, [("input_ids", ("N", "S"), np.int32)])
    assert ids.shape == (4, 1) and ids.dtype == np.int32 and ids.min() >= 0 and ids.max() < 100
    rng = np.random.default_rng(0)
    a, b = rng.standard_normal((5, 8)).astype(np.float32), rng.integers(0, 100, size=(5, 1, 3), dtype=np.int64)
    both = D.generate_random_calibration_data(5, [("a", (None, 8), np.float32), ("b", ("N", "S", 3), np.int64)])
    np.testing.assert_array_equal(both["a"], a)
    np.testing.assert_array_equal(both["b"], b)


def test_multi_input_model_needs_a_dict():
    """calibrate.py:228-233 / test_calibrate.py:229-238."""
    with pytest.raises(ValueError, match="Calibration data must be a dict"):
        D.run_calibration(lambda f: {}, np.zeros((4, 3), np.float32), D.ActivationStream(), input_names=["X", "Y"])


def test_stream_rejects_host_arrays():
    import torch
    with pytest.raises(TypeError, match="not a tensor in GPU memory"):
        D.ActivationStream(input_names=["a"]).feed({"a": torch.zeros(3)})


# ----------------------------------------------------------------------------- GPU
def _mlp(seed=0):
    import torch
    torch.manual_seed(seed)
    m = torch.nn.Sequential()
    m.add_module("fc1", torch.nn.Linear(32, 64, bias=True))
    m.add_module("act", torch.nn.ReLU())
    m.add_module("fc2", torch.nn.Linear(64, 128, bias=False))
    m.add_module("fc3", torch.nn.Linear(128, 32, bias=True))
    return m.cuda()


_TAPS = {"X": ("fc1", "input"), "h1": ("fc1", "output"), "a1": ("fc2", "input"), "h2": ("fc2", "output"),
         "h3": ("fc3", "output")}          # fc3's input IS h2: one value, two roles, like an ONNX graph


def _reference_activations(model, data, batch, num):
    """What calibrate.py:244-251 would hold: a list of per-batch {name: NumPy array} (computed with the same model)."""
    import torch
    runner = D.TorchRunner(model, _TAPS)
    out = []
    for b in O.prepare_calibration_data(data, batch, num):
        out.append({k: v.cpu().numpy() for k, v in runner(torch.from_numpy(b).cuda()).items()})
    runner.close()
    return out


@pytest.mark.gpu
@pytest.mark.parametrize("momentum", [0.0, 0.9])
@pytest.mark.parametrize("kinds", ["input", "output", "both"])
def test_static_activation_parameters_equal_the_hold_everything_flow(momentum, kinds):
    from onnx_quantize_amd.calibration import MinMaxCalibrator
    model = _mlp()
    data = (np.random.default_rng(3).standard_normal((23, 32)) * 2).astype(np.float32)
    in_names = ["X", "a1", "h2"] if kinds != "output" else []
    out_names = ["h1", "h2", "h3"] if kinds != "input" else []
    in_args = QActivationArgs(dtype=QuantType.QUInt8, is_static=True)
    out_args = QActivationArgs(dtype=QuantType.QInt8, symmetric=True, is_static=True)
    stream = D.ActivationStream(calibrator=MinMaxCalibrator(momentum), input_names=in_names, output_names=out_names)
    runner = D.TorchRunner(model, _TAPS)
    D.run_calibration(runner, data, stream, num_samples=23, batch_size=4, input_names=["X"])
    runner.close()
    assert stream.batches == 5                                          # 23 // 4, three samples dropped
    acts = _reference_activations(model, data, 4, 23)
    exp = O.calibrate_flow(acts, in_names, out_names, momentum,
                           ("uint8", False, False) if in_names else None, ("int8", True, False) if out_names else None)
    got = {}
    if in_names:
        got.update({("input", n): v for n, v in stream.input_qparams(in_args).items()})
    if out_names:
        got.update({("output", n): v for n, v in stream.output_qparams(out_args).items()})
    assert got.keys() == exp.keys() and len(got) == len(in_names) + len(out_names)
    for key, (scale, zp) in exp.items():
        s, z = got[key]
        assert s.dtype == np.float32 and s.shape == () and z.dtype == zp.dtype and z.shape == ()
        # bit-exact, EMA and the two-walk EMA of `both` included: an activation zero point is an emitted int8 / uint8
        assert s.tobytes() == np.float32(scale).tobytes() and int(z) == int(zp), (key, float(s), float(scale), int(z), int(zp))


@pytest.mark.gpu
def test_two_walk_ema_differs_from_one_walk_and_is_reproduced():
    """With both kinds and momentum > 0 the reference's output ranges see the batch sequence twice; a driver that
    collected once would give different numbers -- make sure the case is real and the stream follows the reference."""
    import torch
    from onnx_quantize_amd.calibration import MinMaxCalibrator
    rng = np.random.default_rng(5)
    batches = [{"a": rng.standard_normal((4, 8)).astype(np.float32) * (i + 1),
                "b": rng.standard_normal((4, 8)).astype(np.float32)} for i in range(4)]
    args = QActivationArgs(dtype=QuantType.QUInt8, is_static=True)
    stream = D.ActivationStream(calibrator=MinMaxCalibrator(0.5), input_names=["a"], output_names=["b", "a"])
    for b in batches:
        stream.feed({k: torch.from_numpy(v).cuda() for k, v in b.items()})
    exp = O.calibrate_flow(batches, ["a"], ["b", "a"], 0.5, ("uint8", False, False), ("uint8", False, False))
    once = O.calibrate_flow(batches, [], ["b", "a"], 0.5, None, ("uint8", False, False))
    assert exp[("output", "a")][0] != once[("output", "a")][0]
    out, inp = stream.output_qparams(args), stream.input_qparams(args)     # order of the two reads must not matter
    for got, want in ((inp["a"], exp[("input", "a")]), (out["a"], exp[("output", "a")]), (out["b"], exp[("output", "b")])):
        assert got[0].tobytes() == np.float32(want[0]).tobytes() and int(got[1]) == int(want[1])     # bit for bit
    assert not np.isclose(out["a"][0], once[("output", "a")][0], rtol=1e-4)


@pytest.mark.gpu
def test_streamed_hessian_and_gptq_equal_the_concatenated_flow():
    """calibrate.py:288-307 + gptq.py:304-305: H from all batches at once == H streamed batch by batch; then `_gptq`
    with the factor shared by the two layers that read the same value."""
    import torch
    model = _mlp(1)
    data = np.random.default_rng(7).standard_normal((24, 6, 32)).astype(np.float32)       # [samples, seq, K]
    stream = D.ActivationStream(hessian_names=["X", "a1", "h2"], absmax_names=["a1"], keep_names=["a1"])
    runner = D.TorchRunner(model, _TAPS)
    D.run_calibration(runner, data, stream, num_samples=24, batch_size=8)
    runner.close()
    acts = _reference_activations(model, data, 8, 24)
    whole = O.gptq_inputs(acts)
    for name in ("X", "a1", "h2"):
        k = whole[name].shape[-1]
        h, n = O.accumulate_hessian(whole[name], np.zeros((k, k), np.float32), 0)
        assert stream.hessians[name].n == n == 24                         # samples, not tokens (gptq.py:247)
        np.testing.assert_allclose(stream.hessians[name].h.cpu().numpy(), h, rtol=2e-4, atol=2e-5 * np.abs(h).max())
    np.testing.assert_array_equal(stream.absmax["a1"].cpu().numpy(), O.absmax_cols(whole["a1"]))
    np.testing.assert_array_equal(stream.kept("a1").cpu().numpy(), whole["a1"])
    # the weights in the ONNX MatMul layout [K, N]
    layers = {"fc2": (model.fc2.weight.detach().t().contiguous(), "a1"),
              "fc3": (model.fc3.weight.detach().t().contiguous(), "h2")}
    res = D.quantize_weights_gptq(layers, stream.hessians, "int4", "group", 32)
    for name, (w, value) in layers.items():
        q, s, z, _ = res[name]
        eq, es, ez = O.gptq_quantize(w.cpu().numpy(), whole[value], "int4", "group", 32)
        assert q.shape == eq.shape and s.shape == es.shape
        np.testing.assert_allclose(s.cpu().numpy(), es, rtol=1e-5)
        # parity mode: the integers depend on H only through diag(H) == 0 (SURVEY.md finding 1), so a streamed Hessian
        # that differs from the concatenated one in the last bits gives the SAME integers and zero points
        np.testing.assert_array_equal(q.cpu().numpy(), eq)
        np.testing.assert_array_equal(z.cpu().numpy(), ez)
    # inputs of one width are factored as one batch: same results as one factor chain per input
    hs = dict(stream.hessians)
    hs["a1_again"] = stream.hessians["a1"]                                 # a second input of fc2's width
    more = dict(layers, fc2b=(layers["fc2"][0] * 0.5, "a1_again"))
    batched = D.quantize_weights_gptq(more, hs, "int4", "group", 32)
    one_by_one = D.quantize_weights_gptq(more, hs, "int4", "group", 32, factor_batch_bytes=1)
    assert list(batched) == list(more)
    for name in more:
        for a, b in zip(batched[name], one_by_one[name]):
            assert torch.equal(a, b)
    for a, b in zip(batched["fc2"], res["fc2"]):
        assert torch.equal(a, b)
    with pytest.raises(KeyError, match="no Hessian accumulated"):
        D.quantize_weights_gptq({"x": (layers["fc2"][0], "nope")}, stream.hessians, "int4", "group", 32)


@pytest.mark.gpu
def test_stream_reproduces_the_reference_calibrate_model_outputs():
    """The streamed driver against what the reference's `calibrate_model` itself wrote into node.meta
    (tests/golden/calibrate.*): bit for bit, the EMA cases and the two-walk EMA (both kinds) included."""
    import torch
    from conftest import load_json, load_npz
    from onnx_quantize_amd.calibration import MinMaxCalibrator
    G, meta = load_npz("calibrate.npz"), load_json("calibrate.json")
    chain_in, chain_out = ["X", "h1", "h2"], ["h1", "h2", "Y"]
    in_args = QActivationArgs(dtype=QuantType.QUInt8, is_static=True)
    out_args = QActivationArgs(dtype=QuantType.QInt8, symmetric=True, is_static=True)
    for c in meta["cases"]:
        key = c["key"]
        in_names = chain_in if c["kinds"] != "output" else []
        out_names = chain_out if c["kinds"] != "input" else []
        stream = D.ActivationStream(calibrator=MinMaxCalibrator(c["momentum"]), input_names=in_names, output_names=out_names)
        for b in range(c["batches"]):
            stream.feed({n: torch.from_numpy(G[f"{key}_b{b}_{n}"]).cuda() for n in c["names"]})
        got = {}
        if in_names:
            got.update({("input", n): v for n, v in stream.input_qparams(in_args).items()})
        if out_names:
            got.update({("output", n): v for n, v in stream.output_qparams(out_args).items()})
        assert len(got) == len(in_names) + len(out_names)
        for (kind, name), (s, z) in got.items():
            es, ez = G[f"{key}_{kind}_{name}_scale"], G[f"{key}_{kind}_{name}_zp"]
            assert s.tobytes() == es.tobytes() and int(z) == int(ez), (key, kind, name, c["momentum"])   # EMA cases too
            assert z.dtype == ez.dtype
    # the GPTQ branch: streamed Hessians of the concatenated inputs the reference stored
    stream = D.ActivationStream(hessian_names=chain_in)
    for b in range(meta["gptq_batches"]):
        stream.feed({n: torch.from_numpy(G[f"gptq_b{b}_{n}"]).cuda() for n in chain_in})
    for n in chain_in:
        whole = G[f"gptq_input_{n}"]
        h, cnt = O.accumulate_hessian(whole, np.zeros((8, 8), np.float32), 0)
        assert stream.hessians[n].n == cnt == whole.shape[0]
        np.testing.assert_allclose(stream.hessians[n].h.cpu().numpy(), h, rtol=2e-4, atol=2e-5 * np.abs(h).max())
