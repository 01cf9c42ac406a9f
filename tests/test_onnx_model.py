"""SURVEY.md 8f rows N4 / N1 on real ONNX files, without the `onnx` package.

* `onnx_proto`: parse / serialize round trips of files written by ANOTHER producer (torch's C++ exporter over the published
  onnx.proto; tests/golden/make_onnx_fixtures.py) are byte-identical; tensors incl. the 4-bit packing of `core/_pack.py`.
* `graph_runner`: the parsed graphs run to what the torch modules they were exported from compute.
* `onnx_functions`: every `quant`-domain function computes what its reference script says (qfunctions/_qdq/*.py,
  _qlinear/*.py) on plain NumPy restatements of the ONNX operators.
* `model_quantize`: the reference's pipeline (quantize.py:28-80) on the parsed model -- pre-passes, target selection, the
  emission of `emission.plan_node` (pinned node by node on the reference's own rule recordings, test_emission.py), functions,
  opset imports, de-duplication -- with the ORACLE as the numeric provider here; the GPU file runs the same through the
  device path.
"""
import glob
import importlib.util
import os

import numpy as np
import pytest
import torch

import oq_oracle as O
from conftest import ROOT
from onnx_quantize_amd import QActivationArgs, QConfig, QuantType, QWeightArgs, quantize
from onnx_quantize_amd import onnx_proto as P
from onnx_quantize_amd.graph_runner import GraphRunner, UnsupportedOperator
from onnx_quantize_amd.onnx_functions import build_function, function_names

from onnx_model_helpers import FIXTURES, fixture, oracle_weight_arrays, q_oracle


def torch_modules():
    spec = importlib.util.spec_from_file_location("make_onnx_fixtures", os.path.join(ROOT, "tests", "golden", "make_onnx_fixtures.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


# ------------------------------------------------------------------------------------------------------------------ codec
def test_files_of_another_producer_round_trip_byte_for_byte():
    paths = sorted(glob.glob(os.path.join(FIXTURES, "*.onnx")))
    assert len(paths) >= 5
    for path in paths:
        data = open(path, "rb").read()
        model = P.parse_model(data)
        assert P.serialize(model) == data, path
        assert P.serialize(model.copy()) == data, path

        def no_unknown(msg):
            assert not msg._unknown, (path, msg._type)
            for v in msg._values.values():
                for e in (v if isinstance(v, list) else [v]):
                    if isinstance(e, P.Message):
                        no_unknown(e)
        no_unknown(model)                                     # every field torch wrote has a name in the schema table
        assert model.graph.node and model.opset_import[0].version in (17, 21)


def test_wire_primitives():
    t = P.Message("TensorProto", dims=[2, 0, 3], data_type=7, int64_data=[-1, 0, 1 << 40, -(1 << 62)], name="t")
    back = P.parse("TensorProto", P.serialize(t))
    assert back.dims == [2, 0, 3] and back.int64_data == [-1, 0, 1 << 40, -(1 << 62)] and back.name == "t"
    a = P.make_attribute("alpha", 0.25)
    assert P.attribute_value(P.parse("AttributeProto", P.serialize(a))) == 0.25
    for value in (7, -3, "text", [1, -2, 3], [0.5, 1.5], ["a", "b"]):
        got = P.attribute_value(P.parse("AttributeProto", P.serialize(P.make_attribute("v", value))))
        assert got == value, (value, got)
    # a field the table has no name for survives a round trip verbatim, after the known ones
    node = P.make_node("Relu", ["x"], ["y"], name="n")
    raw = P.serialize(node) + bytes([0x98, 0x06, 0x2A])      # key 99 << 3 | 0 as a two-byte varint, then the value 42
    back = P.parse("NodeProto", raw)
    assert back.op_type == "Relu" and len(back._unknown) == 1 and P.serialize(back) == raw
    with pytest.raises(ValueError, match="past the end|truncated"):
        P.parse_model(open(os.path.join(FIXTURES, "mlp_gemm.onnx"), "rb").read()[:-7])
    with pytest.raises(AttributeError):
        node.no_such_field = 1
    # present-but-empty optional fields are kept apart from absent ones
    vi = P.make_value_info("s", P.DataType.FLOAT, [])
    again = P.parse("ValueInfoProto", P.serialize(vi))
    assert again.type.tensor_type.shape is not None and again.type.tensor_type.shape.dim == []
    assert P.parse("ValueInfoProto", P.serialize(P.make_value_info("s", 1, None))).type.tensor_type.shape is None


def test_scalar_fields_sent_with_another_wire_type_are_value_errors():
    """the schema fixes a scalar field's wire type; a float attribute sent as a varint (it used to come back as an int
    and surface later as struct.error in serialize), a float sent as fixed64 or an int64 sent as fixed32 are malformed input."""
    import struct

    def attribute(field, wire, payload):                                    # AttributeProto { name: "a", <field>: payload }
        a = bytearray(b"\x0a\x01a")
        P._put_varint(a, (field << 3) | wire)
        return bytes(a) + payload

    ok = P.parse("AttributeProto", attribute(2, 5, struct.pack("<f", 1.5)))       # f = 1.5, fixed32: the schema's type
    assert ok.f == 1.5
    for field, wire, payload in ((2, 0, b"\x05"),                           # f as varint
                                 (2, 1, struct.pack("<d", 1.5)),             # f as fixed64
                                 (3, 5, struct.pack("<f", 2.0)),             # i (int64) as fixed32
                                 (3, 1, struct.pack("<d", 2.0))):            # i as fixed64
        with pytest.raises(ValueError, match="wire type"):
            P.parse("AttributeProto", attribute(field, wire, payload))


def test_a_half_precision_model_is_refused_by_name_not_returned_untouched():
    """constant fp16 weights under MatMul: `quantize` used to log a warning per node and return the model with a
    success status and nothing rewritten."""
    from onnx_quantize_amd import QConfig, QWeightArgs, quantize
    from onnx_quantize_amd.onnx_proto import DataType, Message, make_node, make_value_info, numpy_to_tensor, serialize

    w = numpy_to_tensor("W", np.ones((8, 4), np.float16))
    assert w.data_type == DataType.FLOAT16
    g = Message("GraphProto", name="half", node=[make_node("MatMul", ["X", "W"], ["Y"], name="fc")], initializer=[w],
                input=[make_value_info("X", DataType.FLOAT16, ["N", 8])], output=[make_value_info("Y", DataType.FLOAT16, ["N", 4])])
    m = Message("ModelProto", ir_version=10, graph=g, opset_import=[Message("OperatorSetIdProto", domain="", version=21)])
    with pytest.raises(NotImplementedError, match="float32"):
        quantize(serialize(m), QConfig(weights=QWeightArgs()))


def test_malformed_files_are_value_errors():
    """A model file is input from outside: whatever a mutated file does to the parser, it ends in ValueError (or parses) --
    no IndexError / struct.error / RecursionError, no hang.  4000 random mutations of a real file + a nesting bomb."""
    import random
    data = open(os.path.join(FIXTURES, "mlp_matmul.onnx"), "rb").read()
    rng = random.Random(0)
    parsed = 0
    for _ in range(4000):
        b = bytearray(data)
        for _ in range(rng.randint(1, 6)):
            op, pos = rng.random(), rng.randrange(len(b))
            if op < 0.5:
                b[pos] = rng.randrange(256)
            elif op < 0.75:
                del b[pos:pos + rng.randint(1, 40)]
            else:
                b[pos:pos] = bytes(rng.randrange(256) for _ in range(rng.randint(1, 12)))
        try:
            m = P.parse_model(bytes(b))
        except ValueError:
            continue
        parsed += 1
        for t in ([] if m.graph is None else m.graph.initializer):
            try:
                P.tensor_to_numpy(t)
            except ValueError:
                pass
        P.serialize(m)
    assert 100 < parsed < 3000
    # graph -> node -> attribute -> graph -> ... 500 levels deep
    inner = b""
    for _ in range(500):
        a = bytearray()
        P._put_varint(a, (6 << 3) | 2)
        P._put_varint(a, len(inner))
        a += inner                                                           # AttributeProto { g: inner }
        n = bytearray()
        P._put_varint(n, (5 << 3) | 2)
        P._put_varint(n, len(a))
        n += a                                                               # NodeProto { attribute: ... }
        gph = bytearray()
        P._put_varint(gph, (1 << 3) | 2)
        P._put_varint(gph, len(n))
        gph += n                                                             # GraphProto { node: ... }
        inner = bytes(gph)
    with pytest.raises(ValueError, match="nested deeper"):
        P.parse("GraphProto", inner)


def test_structural_model_checker():
    """`onnx_proto.check_model`: the structural half of `onnx.checker.check_model`, run on every model the writer emits."""
    for path in sorted(glob.glob(os.path.join(FIXTURES, "*.onnx"))):
        P.check_model(P.load_model(path))
    good = fixture("mlp_matmul")

    def broken(mutate, match):
        m = good.copy()
        mutate(m)
        with pytest.raises(ValueError, match=match):
            P.check_model(m)

    broken(lambda m: setattr(m, "ir_version", None), "ir_version")
    broken(lambda m: setattr(m, "opset_import", []), "default operator set")
    broken(lambda m: m.graph.node.reverse(), "before anything produces it")
    broken(lambda m: setattr(m.graph.node[1], "output", list(m.graph.node[0].output)), "already has a producer")
    broken(lambda m: m.graph.initializer.append(m.graph.initializer[0].copy()), "two initializers")
    broken(lambda m: setattr(m.graph.initializer[0], "dims", [3, 5]), "bytes of raw data")
    broken(lambda m: setattr(m.graph.node[0], "domain", "com.microsoft"), "does not import")
    broken(lambda m: setattr(m.graph.output[0], "name", "nowhere"), "never produced")

    def undefined_call(m):
        m.opset_import.append(P.Message("OperatorSetIdProto", domain="quant", version=1))
        m.graph.node[0].domain, m.graph.node[0].op_type = "quant", "QMatMulWeightsOnlyQDQ"
    broken(undefined_call, "does not define")

    def open_function(m):
        undefined_call(m)
        fn = build_function("QMatMulWeightsOnlyQDQ")
        fn.node[0].input = ["W", "w_scale", "not_an_input"]
        m.functions.append(fn)
    broken(open_function, "function quant::QMatMulWeightsOnlyQDQ.*before anything produces it")


def test_tensors_round_trip_including_four_bit_packing():
    rng = np.random.default_rng(0)
    for dt in (np.float32, np.float16, np.float64, np.int8, np.uint8, np.int32, np.int64, np.bool_):
        a = (rng.standard_normal((3, 5)) * 40).astype(dt)
        t = P.parse("TensorProto", P.serialize(P.numpy_to_tensor("a", a)))
        got = P.tensor_to_numpy(t)
        assert got.dtype == a.dtype and got.shape == a.shape and got.tobytes() == a.tobytes()
    assert P.tensor_to_numpy(P.numpy_to_tensor("s", np.float32(2.5))).shape == ()
    for signed, code in ((True, P.DataType.INT4), (False, P.DataType.UINT4)):
        for shape in ((7,), (3, 5), (4, 6), (1,)):
            lo, hi = (-8, 8) if signed else (0, 16)
            a = rng.integers(lo, hi, size=shape).astype(np.int8 if signed else np.uint8)
            t = P.numpy_to_tensor("q", a, code)
            assert bytes(t.raw_data) == O.pack_nibbles(a).tobytes()          # core/_pack.py:8-22 (pinned by the reference's KATs)
            got = P.tensor_to_numpy(P.parse("TensorProto", P.serialize(t)))
            assert got.dtype == a.dtype and np.array_equal(got, a)
    with pytest.raises(ValueError, match="4-bit range"):
        P.numpy_to_tensor("q", np.array([16], dtype=np.uint8), P.DataType.UINT4)
    # typed fields instead of raw_data (how small tensors are often written)
    t = P.Message("TensorProto", dims=[3], data_type=P.DataType.FLOAT, float_data=[1.0, 2.5, -3.0], name="f")
    assert P.tensor_to_numpy(P.parse("TensorProto", P.serialize(t))).tolist() == [1.0, 2.5, -3.0]
    t = P.Message("TensorProto", dims=[2], data_type=P.DataType.INT8, int32_data=[-5, 7], name="i")
    assert P.tensor_to_numpy(t).tolist() == [-5, 7] and P.tensor_to_numpy(t).dtype == np.int8
    ext = P.Message("TensorProto", dims=[2], data_type=1, name="e", data_location=1)
    with pytest.raises(ValueError, match="external"):
        P.tensor_to_numpy(ext)


def test_external_tensor_data_is_mapped_on_load_and_written_back(tmp_path):
    """Models past protobuf's 2 GiB limit keep their tensors in side files (onnx.proto `external_data`: location / offset /
    length).  They are memory-mapped on load; `save_model(..., external_data=name)` writes them back, large tensors on
    4096-byte boundaries; the in-memory model is left as it was."""
    src = fixture("block")
    inline = P.serialize(src)
    path = tmp_path / "m.onnx"
    P.save_model(src, path, external_data="m.onnx.data", size_threshold=128)
    assert P.serialize(src) == inline                                     # the model in memory is untouched
    assert path.stat().st_size < 12000 and (tmp_path / "m.onnx.data").stat().st_size > 130000
    refs = P.load_model(path, load_external_data=False)
    for t in refs.graph.initializer:
        assert t.data_location == 1 and not t.has("raw_data")
        info = {e.key: e.value for e in t.external_data}
        assert info["location"] == "m.onnx.data" and int(info["offset"]) % 64 == 0
    with pytest.raises(ValueError, match="external"):
        P.tensor_to_numpy(refs.graph.initializer[0])
    back = P.load_model(path)
    assert P.serialize(back) == inline and isinstance(back.graph.initializer[-1].raw_data, memoryview)
    # large tensors are aligned for memory-mapping loaders
    big = P.Message("ModelProto", ir_version=10, graph=P.Message("GraphProto", name="g", initializer=[
        P.numpy_to_tensor("small", np.arange(300, dtype=np.float32)), P.numpy_to_tensor("big", np.ones((512, 1024), dtype=np.float32))]))
    P.save_model(big, tmp_path / "b.onnx", external_data="b.data")
    offs = {t.name: int({e.key: e.value for e in t.external_data}["offset"]) for t in P.load_model(tmp_path / "b.onnx", False).graph.initializer}
    assert offs == {"small": 0, "big": 4096}
    assert np.array_equal(P.tensor_to_numpy(P.load_model(tmp_path / "b.onnx").graph.initializer[1]), np.ones((512, 1024), dtype=np.float32))
    # references that leave the model's directory or the file are refused
    evil = P.load_model(path, load_external_data=False)
    evil.graph.initializer[0].external_data[0].value = "../m.onnx.data"
    with pytest.raises(ValueError, match="leaves the model's directory"):
        P.resolve_external_data(evil, tmp_path)
    outside = tmp_path.parent / f"{tmp_path.name}_outside.data"
    outside.write_bytes(b"\0" * 4096)
    (tmp_path / "link.data").symlink_to(outside)                          # a link out of the directory is a way out of it
    evil = P.load_model(path, load_external_data=False)
    evil.graph.initializer[0].external_data[0].value = "link.data"
    with pytest.raises(ValueError, match="leaves the model's directory"):
        P.resolve_external_data(evil, tmp_path)
    evil = P.load_model(path, load_external_data=False)
    next(e for e in evil.graph.initializer[0].external_data if e.key == "length").value = str(1 << 40)
    with pytest.raises(ValueError, match="outside"):
        P.resolve_external_data(evil, tmp_path)
    with pytest.raises(ValueError, match="file name"):
        P.save_model(src, path, external_data="sub/m.data")
    # quantize_file: a source with side files gives a result with a side file; the quantized tensors are the inline run's
    import onnx_quantize_amd.model_quantize as MQ
    qc = CONFIGS["uint4_g32"]()
    out = MQ.quantize_file(path, tmp_path / "q.onnx", qc, weight_arrays=oracle_weight_arrays, quantize_bias=O.quantize_bias)
    assert (tmp_path / "q.onnx.data").exists()
    assert P.serialize(P.load_model(tmp_path / "q.onnx")) == P.serialize(out) == P.serialize(q_oracle(fixture("block"), qc))
    MQ.quantize_file(os.path.join(FIXTURES, "block.onnx"), tmp_path / "q2.onnx", qc, weight_arrays=oracle_weight_arrays, quantize_bias=O.quantize_bias)
    assert not (tmp_path / "q2.onnx.data").exists() and (tmp_path / "q2.onnx").read_bytes() == P.serialize(out)


# ------------------------------------------------------------------------------------------------------------------ runner
def test_graph_runner_matches_the_exported_torch_modules():
    fx = torch_modules()
    torch.manual_seed(0)
    mlp = torch.nn.Sequential(torch.nn.Linear(64, 48), torch.nn.ReLU(), torch.nn.Linear(48, 16, bias=False)).eval()
    torch.manual_seed(1)
    block = fx.Block().eval()
    torch.manual_seed(2)
    tied = fx.Tied().eval()
    gen = torch.Generator().manual_seed(5)
    cases = [("mlp_gemm", mlp, torch.randn(7, 64, generator=gen), "y"), ("mlp_matmul", mlp, torch.randn(3, 4, 64, generator=gen), "y"),
             ("block", block, torch.randn(3, 6, 64, generator=gen), "y"), ("tied", tied, torch.randint(0, 40, (2, 7), generator=gen), "logits")]
    with torch.no_grad():
        for name, module, x, out in cases:
            got = GraphRunner(fixture(name), device="cpu")(x)[out]
            torch.testing.assert_close(got, module(x), rtol=1e-5, atol=1e-5)
    # taps: only what the wanted values need is run, and model inputs can be asked for
    m = fixture("block")
    r = GraphRunner(m, outputs=["/ln1/LayerNormalization_output_0", "x"], device="cpu")
    assert len(r.nodes) == 1 and set(r(cases[2][2])) == {"/ln1/LayerNormalization_output_0", "x"}
    # a sink gets every wanted value the moment it exists (same tensors as the dictionary form), nothing is returned
    taps = ["x", "/ln1/LayerNormalization_output_0", "/up/MatMul_output_0", "y"]
    whole = GraphRunner(m, outputs=taps, device="cpu")(cases[2][2])
    seen = {}
    assert GraphRunner(m, outputs=taps, device="cpu")(cases[2][2], sink=lambda name, t: seen.__setitem__(name, t.clone())) == {}
    assert list(seen) == ["x", "/ln1/LayerNormalization_output_0", "/up/MatMul_output_0", "y"] and all(torch.equal(seen[k], whole[k]) for k in taps)
    with pytest.raises(KeyError, match="no value named"):
        GraphRunner(m, outputs=["nope"], device="cpu")
    bad = m.copy()
    bad.graph.node[11].op_type = "NoSuchOp"
    with pytest.raises(UnsupportedOperator, match="NoSuchOp"):
        GraphRunner(bad, device="cpu")(cases[2][2])


# ------------------------------------------------------------------------------------------------------------------ functions
def _np_dq(q, s, z, axis=1):
    q, s, z = np.asarray(q, np.float32), np.asarray(s, np.float32), np.asarray(z, np.float32)
    if s.ndim == 1 and q.ndim > 1:
        shape = [1] * q.ndim
        shape[axis] = -1
        s, z = s.reshape(shape), z.reshape(shape)
    return (q - z) * s


def _np_q(x, s, z):
    lo, hi = (0, 255) if z.dtype == np.uint8 else (-128, 127)
    return np.clip(np.rint(x / s) + z.astype(np.float32), lo, hi).astype(z.dtype)


def _np_dynq(x):
    lo, hi = min(float(x.min()), 0.0), max(float(x.max()), 0.0)
    s = np.float32((hi - lo) / 255.0)
    z = np.uint8(np.rint(np.clip(-lo / s, 0, 255)))
    return np.clip(np.rint(x / s) + np.float32(z), 0, 255).astype(np.uint8), s, z


def _declared(names, inputs):
    """graph inputs typed like the data that will be fed (the runner refuses a feed of another type, like a session)."""
    code = lambda v: P._CODE_OF[np.dtype(v.numpy().dtype if isinstance(v, torch.Tensor) else np.asarray(v).dtype)]      # noqa: E731
    return [P.make_value_info(n, code(v), None) for n, v in zip(names, inputs)]


def _call(fn, inputs, attrs=None):
    """Run ONE call of `fn` through the graph runner: a model whose only node is the call."""
    names = [f"i{k}" for k in range(len(inputs))]
    node = P.make_node(fn.name, names, ["out"], domain="quant", **(attrs or {}))
    g = P.Message("GraphProto", node=[node], name="g", input=_declared(names, inputs), output=[P.make_value_info("out", 1, None)])
    model = P.Message("ModelProto", ir_version=10, graph=g, functions=[fn],
                      opset_import=[P.Message("OperatorSetIdProto", domain="", version=21), P.Message("OperatorSetIdProto", domain="quant", version=1),
                                    P.Message("OperatorSetIdProto", domain="com.microsoft", version=1)])
    model = P.parse_model(P.serialize(model))                 # through the wire, like a file
    return GraphRunner(model, device="cpu")(dict(zip(names, inputs)))["out"].numpy()


def _check_function(fn):
    """What `onnx.checker.check_function` checks of a FunctionProto's structure (qfunctions tests call it on every function):
    a name and a domain, unique inputs / outputs, nodes in SSA form that only read inputs or earlier outputs (a function
    body is closed), every output produced, every operator domain imported."""
    assert fn.name and fn.domain == "quant"
    assert len(set(fn.input)) == len(fn.input) and len(set(fn.output)) == len(fn.output)
    known, written = set(fn.input), set()
    imported = {o.domain or "" for o in fn.opset_import}
    for n in fn.node:
        assert n.op_type and (n.domain or "") in imported, (fn.name, n.op_type, n.domain)
        for v in n.input:
            assert v == "" or v in known, (fn.name, n.op_type, v)
        for o in n.output:
            assert o and o not in written and o not in fn.input, (fn.name, o)
            written.add(o)
            known.add(o)
    assert all(o in written for o in fn.output), fn.name
    assert {o.domain or "": o.version for o in fn.opset_import}[""] == 21            # opset.py:4


def test_function_protos_are_well_formed():
    for name in function_names():
        grouped = "Grouped" in name
        for four_bit in ((False, True) if grouped else (False,)):
            _check_function(build_function(name, group_size=8 if grouped else None, four_bit=four_bit))


def test_every_quant_function_computes_its_reference_script():
    rng = np.random.default_rng(3)
    k, n = 32, 12
    x = rng.standard_normal((5, k)).astype(np.float32)
    w = (rng.standard_normal((k, n)) * 0.1).astype(np.float32)
    bias = (rng.standard_normal(n) * 0.05).astype(np.float32)
    wq, ws, wz = O.rtn_quantize(w, "int8", "channel", -1, False)
    bq, bs, bz = O.rtn_quantize(bias.reshape(1, -1), "int8", "tensor", -1, False)
    bq = bq.reshape(-1)
    xs, xz = np.float32(0.03), np.uint8(120)
    os_, oz = np.float32(0.05), np.uint8(131)
    wd, bd = _np_dq(wq, ws, wz), _np_dq(bq, bs, bz)
    seen = set()
    for name in function_names():
        if "Grouped" in name or name.startswith("QLinear"):
            continue
        gemm = name.startswith("QGemm")
        fn = build_function(name)
        static_in, static_out = "x_scale" in fn.input, "out_scale" in fn.input
        dyn_in = "DynamicInput" in name
        dyn_out = "DynamicOutput" in name or "DynamicInputOutput" in name
        args = [x, wq] + ([bq] if gemm else []) + [ws, wz] + ([bs, bz] if gemm else [])
        args += ([xs, xz] if static_in else []) + ([os_, oz] if static_out else [])
        assert len(args) == len(fn.input), (name, list(fn.input))
        xin = x
        if static_in:
            xin = _np_dq(_np_q(x, xs, xz), xs, xz)
        elif dyn_in:
            q, s, z = _np_dynq(x)
            xin = _np_dq(q, s, z)
        want = xin @ wd + (bd if gemm else 0)
        if static_out:
            want = _np_dq(_np_q(want, os_, oz), os_, oz)
        elif dyn_out:
            q, s, z = _np_dynq(want)
            want = _np_dq(q, s, z)
        got = _call(fn, args)
        # a value on a rounding boundary may fall either way between NumPy and torch: one quantization step of slack
        np.testing.assert_allclose(got, want, rtol=1e-5, atol=(float(os_) if (static_out or dyn_out) else 1e-5) + 1e-5, err_msg=name)
        seen.add(name)
    assert len(seen) == 14
    # grouped: W [K, N] int8 | int4 / uint4 containers, parameters [N*K/g, 1] (rtn.py:106-109)
    for qtype in ("int8", "int4", "uint4"):
        g = 8
        q, s, z = O.rtn_quantize(w, qtype, "group", g, False)
        wd_g = O.dequantize(q, s, z, rows_of="group", group_size=g)
        shape_t = np.asarray(w.T.shape, dtype=np.int64)
        for op in ("MatMul", "Gemm"):
            fn = build_function(f"Q{op}WeightsOnlyGrouped", group_size=g, four_bit=qtype != "int8")
            args = [x, q] + ([bq] if op == "Gemm" else []) + [s, z] + ([bs, bz] if op == "Gemm" else []) + [shape_t]
            got = _call(fn, args, {"num_bits": 8})
            np.testing.assert_allclose(got, x @ wd_g + (bd if op == "Gemm" else 0), rtol=1e-5, atol=1e-5)
    # QLinear: integer operators between Q and DQ
    wq8, ws8, wz8 = O.rtn_quantize(w, "int8", "tensor", -1, True)
    acc = (_np_q(x, xs, xz).astype(np.float32) - np.float32(xz)) * xs @ _np_dq(wq8, ws8, wz8)
    got = _call(build_function("QLinearMatMul"), [x, wq8, ws8, wz8, xs, xz, os_, oz])
    np.testing.assert_allclose(got, _np_dq(_np_q(acc, os_, oz), os_, oz), rtol=1e-5, atol=float(os_) + 1e-5)
    b32, _, _ = O.quantize_bias(bias, xs, ws8)
    got = _call(build_function("QLinearGemm"), [x, wq8, b32, ws8, wz8, xs, xz, os_, oz])
    acc_b = acc + b32.astype(np.float32) * (xs * ws8)
    np.testing.assert_allclose(got, _np_dq(_np_q(acc_b, os_, oz), os_, oz), rtol=1e-5, atol=float(os_) + 1e-5)
    with pytest.raises(KeyError):
        build_function("QNothing")
    with pytest.raises(ValueError, match="group size"):
        build_function("QMatMulWeightsOnlyGrouped")


# ------------------------------------------------------------------------------------------------------------------ pipeline
CONFIGS = {
    "int8_tensor_sym": lambda: QConfig(weights=QWeightArgs(dtype=QuantType.QInt8, symmetric=True)),          # BASELINE config 1
    "uint4_g32": lambda: QConfig(weights=QWeightArgs(dtype=QuantType.QUInt4, group_size=32)),                # config 2's shape of rule
    "int4_g32": lambda: QConfig(weights=QWeightArgs(dtype=QuantType.QInt4, group_size=32)),
    "int8_channel": lambda: QConfig(weights=QWeightArgs(dtype=QuantType.QInt8, group_size=-1)),
    "uint8_g16": lambda: QConfig(weights=QWeightArgs(dtype=QuantType.QUInt8, group_size=16)),
}


def _weights_of(model):
    """{node name: (op, weight [K, N] with transB applied, bias | None)} of the float model's MatMul / Gemm nodes."""
    inits = {t.name: t for t in model.graph.initializer}
    out = {}
    for n in model.graph.node:
        if n.op_type in ("MatMul", "Gemm") and len(n.input) > 1 and n.input[1] in inits:
            w = P.tensor_to_numpy(inits[n.input[1]])
            if n.op_type == "Gemm" and any(a.name == "transB" and a.i for a in n.attribute):
                w = w.T
            b = P.tensor_to_numpy(inits[n.input[2]]) if len(n.input) > 2 and n.input[2] in inits else None
            out[n.name] = (n.op_type, np.ascontiguousarray(w), b)
    return out


@pytest.mark.parametrize("cfg", sorted(CONFIGS))
@pytest.mark.parametrize("name", ["mlp_gemm", "mlp_matmul", "block", "wide_matmul", "tied"])
def test_quantized_model_structure_and_numbers(name, cfg):
    src = fixture(name)
    before = P.serialize(src)
    qc = CONFIGS[cfg]()
    a = qc.weights
    out = P.parse_model(P.serialize(q_oracle(src, qc)))       # through the wire
    assert P.serialize(src) == before                         # the caller's model is not touched (ir.from_proto copies)
    floats = _weights_of(src)
    inits = {t.name: t for t in out.graph.initializer}
    assert len(inits) == len(out.graph.initializer)           # unique names
    by_name = {n.name: n for n in out.graph.node}
    assert {o.domain or "": o.version for o in out.opset_import}[""] == 21 and out.ir_version >= 10
    used_domains, used_fns = set(), set()
    for node_name, (op, w, b) in floats.items():
        n = by_name[node_name]
        k = w.shape[0]
        g = O.resolve_group_size(k, a.group_size) if a.strategy.value == "group" else a.group_size
        nbits = O.matmul_nbits_compatible(a.dtype.key, a.strategy.value, g)
        wname = n.input[1]
        q, s, z = O.seam_arrays(w, "rtn", a.dtype.key, a.strategy.value, g, a.symmetric, a.reduce_range, a.clip_ratio, a.mse, nbits=nbits)
        # parameters by position in the call: byte-identical initializers of different weights are ONE initializer after
        # DeduplicateInitializersPass (quantize.py:75), e.g. the 0-d zero point 0 of every symmetric weight
        gemm_call = n.op_type.startswith("QGemm")
        sname, zname = (n.input[3], n.input[4]) if gemm_call else (n.input[2], n.input[3])
        got_q, got_s, got_z = (P.tensor_to_numpy(inits[v]) for v in (wname, sname, zname))
        assert np.array_equal(got_q, q) and got_s.tobytes() == np.asarray(s).tobytes() and np.array_equal(got_z, z), (node_name, cfg)
        if nbits:
            assert (n.op_type, n.domain) == ("MatMulNBits", "com.microsoft")
            attrs = {x.name: P.attribute_value(x) for x in n.attribute}
            assert attrs == dict(K=k, N=w.shape[1], bits=a.dtype.bitwidth, block_size=g)
            assert list(n.input)[:5] == [n.input[0], wname, sname, zname, ""]
            assert inits[wname].data_type == P.DataType.UINT8
            if b is not None:
                assert len(n.input) == 6 and inits[n.input[5]].data_type == P.DataType.FLOAT        # the bias stays float
        else:
            assert n.domain == "quant"
            want_code = {"int8": P.DataType.INT8, "uint8": P.DataType.UINT8, "int4": P.DataType.INT4, "uint4": P.DataType.UINT4}[a.dtype.key]
            assert inits[wname].data_type == want_code and inits[zname].data_type == want_code
            family = "Gemm" if (op == "Gemm" and b is not None) else "MatMul"
            grouped = a.strategy.value == "group"
            assert n.op_type == f"Q{family}WeightsOnly" + ("Grouped" if grouped else "QDQ")
            assert {x.name: P.attribute_value(x) for x in n.attribute} == {"num_bits": a.dtype.bitwidth}
            if grouped:
                assert P.tensor_to_numpy(inits[n.input[-1]]).tolist() == [w.shape[1], w.shape[0]]
            if family == "Gemm":
                # the bias config passes `is_symmetric=`, a keyword QWeightArgs ignores (gemm_to_qgemm.py:47-62): asymmetric always
                bq, bs, bz = O.rtn_quantize(b.reshape(1, -1), a.dtype.key, "tensor", -1, False, a.reduce_range, a.clip_ratio)
                assert np.array_equal(P.tensor_to_numpy(inits[n.input[2]]), bq.reshape(-1))
                assert inits[n.input[2]].data_type == want_code
            used_fns.add((n.op_type, n.overload or ""))
        used_domains.add(n.domain)
    # functions: exactly the ones the calls use; opset imports: exactly the domains in use
    assert {(f.name, f.overload or "") for f in out.functions} == used_fns
    assert all(f.domain == "quant" for f in out.functions)
    assert {o.domain for o in out.opset_import if o.domain} == used_domains
    # no leftovers of the float weights, and nothing dangling
    produced = {o for n in out.graph.node for o in n.output} | set(inits) | {i.name for i in out.graph.input}
    assert all(v in produced for n in out.graph.node for v in n.input if v)
    assert not [n.name for n in out.graph.node if n.domain and inits[n.input[1]].data_type == P.DataType.FLOAT]
    # the quantized model computes the float model with every weight replaced by its dequantized value
    gen = torch.Generator().manual_seed(11)
    feed = {"mlp_gemm": torch.randn(5, 64, generator=gen), "mlp_matmul": torch.randn(2, 3, 64, generator=gen),
            "block": torch.randn(2, 6, 64, generator=gen), "wide_matmul": torch.randn(2, 3, 256, generator=gen),
            "tied": torch.randint(0, 40, (2, 7), generator=gen)}[name]
    fake = src.copy()
    for n in list(fake.graph.node):
        if n.name in floats:
            op, w, b = floats[n.name]
            g = O.resolve_group_size(w.shape[0], a.group_size) if a.strategy.value == "group" else a.group_size
            q, s, z = O.rtn_quantize(w, a.dtype.key, a.strategy.value, g if g is not None else -1, a.symmetric, a.reduce_range, a.clip_ratio)
            wd = O.dequantize(q, s, z, rows_of=a.strategy.value, group_size=g if g is not None else -1).astype(np.float32)
            transposed = n.op_type == "Gemm" and any(x.name == "transB" and x.i for x in n.attribute)
            new_name = f"{n.input[1]}@{n.name}"                     # tied weights: each consumer has its own dequantized copy
            fake.graph.initializer.append(P.numpy_to_tensor(new_name, np.ascontiguousarray(wd.T if transposed else wd)))
            ins = list(n.input)
            ins[1] = new_name
            nbits = O.matmul_nbits_compatible(a.dtype.key, a.strategy.value, g)
            if op == "Gemm" and b is not None and not nbits:
                bq, bs, bz = O.rtn_quantize(b.reshape(1, -1), a.dtype.key, "tensor", -1, False, a.reduce_range, a.clip_ratio)
                bname = f"{n.input[2]}@{n.name}"
                fake.graph.initializer.append(P.numpy_to_tensor(bname, ((bq.astype(np.float32) - np.float32(bz)) * bs).reshape(-1).astype(np.float32)))
                ins[2] = bname
            n.input = ins
    want = GraphRunner(fake, device="cpu")(feed)
    got = GraphRunner(out, device="cpu")(feed)
    for key in want:
        torch.testing.assert_close(got[key], want[key], rtol=2e-5, atol=2e-5)


def test_pre_passes_follow_the_reference():
    # StandarizeGemm: transB = 1 becomes transB = 0 with the weight transposed; the re-emitted node carries transB only
    src = fixture("mlp_gemm")
    w_before = P.tensor_to_numpy(next(t for t in src.graph.initializer if t.name == "0.weight"))
    assert w_before.shape == (48, 64)
    out = q_oracle(src, CONFIGS["int8_channel"]())
    gemm = out.graph.node[0]
    assert gemm.op_type == "QGemmWeightsOnlyQDQ" and P.tensor_to_numpy(next(t for t in out.graph.initializer if t.name == "0.weight")).shape == (64, 48)
    # DuplicateInitializersPass: the second consumer of a tied matrix gets `<name>_1` and is quantized on its own ...
    import onnx_quantize_amd.model_quantize as MQ
    tied = MQ.as_model(fixture("tied"))
    MQ._duplicate_shared_initializers(MQ._Graph(tied.graph))
    assert [n.input[1] for n in tied.graph.node if n.op_type == "MatMul"] == ["w", "w_1"]
    assert [t.name for t in tied.graph.initializer] == ["w", "emb.weight", "w_1"]
    # ... and DeduplicateInitializersPass folds what is byte-identical afterwards: same weight -> same q, scale and zero point
    out = q_oracle(fixture("tied"), CONFIGS["int8_tensor_sym"]())
    mm = [n for n in out.graph.node if n.domain == "quant"]
    names = [t.name for t in out.graph.initializer]
    assert "w" in names and "w_1" not in names and "w_1/scale" not in names
    assert [list(n.input[1:]) for n in mm] == [["w", "w/scale", "w/zero_point"]] * 2
    # `ignore`: regular expressions searched in node names (calibrate.py:63-70)
    qc = QConfig(weights=QWeightArgs(dtype=QuantType.QInt8), ignore=[r"/(q|k)/MatMul$", "down"])
    out = q_oracle(fixture("block"), qc)
    kept = {n.name for n in out.graph.node if n.op_type == "MatMul"}
    assert {"/q/MatMul", "/k/MatMul", "/down/MatMul"} <= kept and "/v/MatMul" not in kept
    # target_op_types: Gemm only
    out = q_oracle(fixture("mlp_gemm"), QConfig(weights=QWeightArgs(dtype=QuantType.QInt8), target_op_types=["Gemm"]))
    assert [n.op_type for n in out.graph.node] == ["QGemmWeightsOnlyQDQ", "Relu", "MatMul"]
    # MatMul + Add on matrices -> Gemm (onnxscript's matmul_add_to_gemm_rule), then quantized as a Gemm with bias
    x2 = P.make_value_info("x", 1, ["batch", 64])
    w = np.random.default_rng(0).standard_normal((64, 8)).astype(np.float32)
    b = np.linspace(-1, 1, 8, dtype=np.float32)
    g = P.Message("GraphProto", name="g", input=[x2], output=[P.make_value_info("y", 1, ["batch", 8])],
                  node=[P.make_node("MatMul", ["x", "w"], ["mm"], name="mm"), P.make_node("Add", ["mm", "b"], ["y"], name="add")],
                  initializer=[P.numpy_to_tensor("w", w), P.numpy_to_tensor("b", b)])
    model = P.Message("ModelProto", ir_version=8, graph=g, opset_import=[P.Message("OperatorSetIdProto", domain="", version=17)])
    out = q_oracle(model, CONFIGS["int8_channel"]())
    assert [n.op_type for n in out.graph.node] == ["QGemmWeightsOnlyQDQ"] and list(out.graph.node[0].input)[:3] == ["x", "w", "b"]
    x = torch.randn(3, 64)
    torch.testing.assert_close(GraphRunner(out, device="cpu")(x)["y"], GraphRunner(model, device="cpu")(x)["y"], rtol=0.05, atol=0.05)
    # rank-3 input: the rule does not apply, MatMul and Add stay apart
    g.input = [P.make_value_info("x", 1, ["batch", 4, 64])]
    out = q_oracle(model, CONFIGS["int8_channel"]())
    assert [n.op_type for n in out.graph.node] == ["QMatMulWeightsOnlyQDQ", "Add"]


def test_weights_produced_by_constant_nodes_are_quantized_too():
    """`ir.convenience.get_const_tensor` accepts a value a Constant node produces (calibrate.py:75-85, matmul_to_qmatmul.py:38)."""
    w = np.random.default_rng(2).standard_normal((16, 8)).astype(np.float32)
    b = np.linspace(-1, 1, 8, dtype=np.float32)
    g = P.Message("GraphProto", name="g", input=[P.make_value_info("x", 1, ["n", 16])], output=[P.make_value_info("y", 1, None)],
                  node=[P.make_node("Constant", [], ["w"], name="cw", value=P.numpy_to_tensor("", w)),
                        P.make_node("Constant", [], ["b"], name="cb", value=P.numpy_to_tensor("", b)),
                        P.make_node("Gemm", ["x", "w", "b"], ["y"], name="fc")])
    model = P.Message("ModelProto", ir_version=10, graph=g, opset_import=[P.Message("OperatorSetIdProto", domain="", version=21)])
    out = q_oracle(model, CONFIGS["int8_channel"]())
    assert [n.op_type for n in out.graph.node] == ["QGemmWeightsOnlyQDQ"]
    q, s, z = O.rtn_quantize(w, "int8", "channel", -1, False)
    inits = {t.name: P.tensor_to_numpy(t) for t in out.graph.initializer}
    assert np.array_equal(inits["w"], q) and inits["w/scale"].tobytes() == s.tobytes()


def test_constants_are_folded_like_the_optimizer_does():
    """quantize.py:52 (`onnxscript.optimizer.optimize`) decides which weights are constants by the time the rules look: Identity
    nodes are gone, `MatMul(x, Transpose(W))` reads a folded initializer (Transpose folds at any size), other constant
    expressions fold only under the folder's size limits (8192 input elements, 512 * 512 output elements)."""
    rng = np.random.default_rng(6)
    wt = rng.standard_normal((96, 128)).astype(np.float32)                   # stored [N, K], 12288 elements > 8192
    zeros = np.zeros(96, dtype=np.float32)
    small, big = rng.standard_normal((16, 96)).astype(np.float32), rng.standard_normal((96, 100)).astype(np.float32)
    g = P.Message("GraphProto", name="g", input=[P.make_value_info("x", 1, ["n", 128])], output=[P.make_value_info("y", 1, None), P.make_value_info("z", 1, None)],
                  node=[P.make_node("Transpose", ["wt"], ["w"], name="t", perm=[1, 0]),
                        P.make_node("Identity", ["zeros"], ["bias"], name="alias"),
                        P.make_node("Gemm", ["x", "w", "bias"], ["h"], name="fc", transB=0),
                        P.make_node("Mul", ["small", "two"], ["small2"], name="scale_small"),          # 1536 elements: folds
                        P.make_node("Mul", ["big", "two"], ["big2"], name="scale_big"),                # 9600 elements: stays
                        P.make_node("MatMul", ["h", "small2t"], ["y"], name="fc_small"),
                        P.make_node("MatMul", ["h", "big2"], ["z"], name="fc_big")],
                  initializer=[P.numpy_to_tensor("wt", wt), P.numpy_to_tensor("zeros", zeros), P.numpy_to_tensor("small", small),
                               P.numpy_to_tensor("big", big), P.numpy_to_tensor("two", np.float32(2.0)), P.numpy_to_tensor("small2t", small.T.copy() * 2)])
    model = P.Message("ModelProto", ir_version=10, graph=g, opset_import=[P.Message("OperatorSetIdProto", domain="", version=21)])
    out = q_oracle(model, CONFIGS["int8_channel"]())
    ops = {n.name: (n.op_type, n.domain or "") for n in out.graph.node}
    assert "t" not in ops and "alias" not in ops and "scale_small" not in ops           # folded / eliminated (scale_small: unused after folding)
    assert ops["fc"] == ("QGemmWeightsOnlyQDQ", "quant")                                # weight from the folded Transpose, bias through the alias
    assert ops["fc_small"][1] == "quant"
    assert ops["scale_big"] == ("Mul", "") and ops["fc_big"] == ("MatMul", "")          # over the input limit: not a constant weight
    inits = {t.name: P.tensor_to_numpy(t) for t in out.graph.initializer}
    q, s, z = O.rtn_quantize(np.ascontiguousarray(wt.T), "int8", "channel", -1, False)
    assert np.array_equal(inits["w"], q) and inits["w/scale"].tobytes() == s.tobytes()
    x = torch.randn(4, 128)
    for k, v in GraphRunner(model, device="cpu")(x).items():
        got = GraphRunner(out, device="cpu")(x)[k]
        assert ((got - v).norm() / v.norm()).item() < 0.03


def test_target_nodes_lists_what_would_be_quantized_without_a_gpu():
    from onnx_quantize_amd.model_quantize import target_nodes
    src = fixture("block")
    before = P.serialize(src)
    got = target_nodes(src, QConfig(weights=QWeightArgs(dtype=QuantType.QInt8), ignore=["down"]))
    assert [t[0] for t in got] == ["/q/MatMul", "/k/MatMul", "/v/MatMul", "/o/MatMul", "/up/MatMul"] and got[-1][1:] == ("MatMul", "onnx::MatMul_111", [64, 128])
    assert P.serialize(src) == before
    assert [t[:2] for t in target_nodes(fixture("mlp_gemm"), QConfig(weights=QWeightArgs(), target_op_types=["Gemm"]))] == [("/0/Gemm", "Gemm")]
    assert target_nodes(fixture("mlp_gemm"), QConfig(weights=QWeightArgs()))[0][3] == [64, 48]      # transB = 1 standardised: [K, N]


def test_opset_is_raised_with_adapters_or_refused_by_name():
    x = P.make_value_info("x", 1, ["batch", 16])
    w = np.random.default_rng(1).standard_normal((16, 4)).astype(np.float32)

    def model(extra_nodes, opset, more_inputs=()):
        g = P.Message("GraphProto", name="g", input=[x] + [P.make_value_info(v, 1, None) for v in more_inputs], output=[P.make_value_info("y", 1, None)],
                      node=[P.make_node("MatMul", ["x", "w"], ["h"], name="fc")] + extra_nodes, initializer=[P.numpy_to_tensor("w", w)])
        return P.Message("ModelProto", ir_version=7, graph=g, opset_import=[P.Message("OperatorSetIdProto", domain="", version=opset)])

    m = model([P.make_node("ReduceMean", ["h"], ["y"], name="rm", axes=[-1], keepdims=0)], 13)
    out = q_oracle(m, CONFIGS["int8_channel"]())
    rm = out.graph.node[1]
    assert rm.op_type == "ReduceMean" and len(rm.input) == 2 and [a.name for a in rm.attribute] == ["keepdims"]
    assert P.tensor_to_numpy(next(t for t in out.graph.initializer if t.name == rm.input[1])).tolist() == [-1]
    xin = torch.randn(5, 16)
    torch.testing.assert_close(GraphRunner(out, device="cpu")(xin)["y"], GraphRunner(m, device="cpu")(xin)["y"], rtol=0.05, atol=0.05)
    m = model([P.make_node("Split", ["h"], ["y", "y2"], name="sp", axis=1)], 13)
    out = q_oracle(m, CONFIGS["int8_channel"]())
    assert {a.name: P.attribute_value(a) for a in out.graph.node[1].attribute} == {"axis": 1, "num_outputs": 2}
    with pytest.raises(NotImplementedError, match="GroupNormalization"):
        q_oracle(model([P.make_node("GroupNormalization", ["h", "s", "b"], ["y"], name="gn", num_groups=2)], 18, ("s", "b")), CONFIGS["int8_channel"]())
    with pytest.raises(NotImplementedError, match="BatchNormalization"):
        q_oracle(model([P.make_node("BatchNormalization", ["h", "s", "b", "m", "v"], ["y", "m2", "v2"], name="bn")], 13, ("s", "b", "m", "v")),
                 CONFIGS["int8_channel"]())
    # the converter's other adapters between 13 and 21: an attribute made explicit, renamed, or moved to an input
    out = q_oracle(model([P.make_node("RoiAlign", ["h", "rois", "idx"], ["y"], name="roi")], 15, ("rois", "idx")), CONFIGS["int8_channel"]())
    assert {a.name: P.attribute_value(a) for a in out.graph.node[1].attribute} == {"coordinate_transformation_mode": "output_half_pixel"}
    nodes = [P.make_node("GridSample", ["h", "grid"], ["gs"], name="gs", mode="bilinear"), P.make_node("DFT", ["gs"], ["y"], name="dft", axis=2, onesided=1)]
    out = q_oracle(model(nodes, 17, ("grid",)), CONFIGS["int8_channel"]())
    gs, dft = out.graph.node[1:3]
    assert {a.name: P.attribute_value(a) for a in gs.attribute} == {"mode": "linear"}
    assert list(dft.input)[:2] == ["gs", ""] and [a.name for a in dft.attribute] == ["onesided"]
    axis = next(t for t in out.graph.initializer if t.name == dft.input[2])
    assert list(axis.dims) == [] and axis.data_type == P.DataType.INT64 and int(P.tensor_to_numpy(axis)) == 2
    out = q_oracle(model([P.make_node("RoiAlign", ["h", "rois", "idx"], ["y"], name="roi", coordinate_transformation_mode="half_pixel")], 16, ("rois", "idx")),
                   CONFIGS["int8_channel"]())
    assert {a.name: P.attribute_value(a) for a in out.graph.node[1].attribute} == {"coordinate_transformation_mode": "half_pixel"}
    with pytest.raises(NotImplementedError, match="opset 11"):
        q_oracle(model([P.make_node("Relu", ["h"], ["y"], name="r")], 11), CONFIGS["int8_channel"]())
    out = q_oracle(model([P.make_node("Relu", ["h"], ["y"], name="r")], 22), CONFIGS["int8_channel"]())
    assert out.opset_import[0].version == 22                  # newer models keep their opset


def test_mixed_group_sizes_stay_correct_as_overloads():
    """qmatmul.py:218-236 registers one function per group size under ONE name and qfunctions/__init__.py:17-20 keeps the
    last: a model whose weights resolve to different group sizes (base.py:72) would call it with the wrong block size.  The
    writer emits one overload per size."""
    out = q_oracle(fixture("mlp_matmul"), CONFIGS["int4_g32"]())          # K = 64 -> g 32; K = 48 -> g 48
    fns = {(f.name, f.overload) for f in out.functions}
    assert fns == {("QMatMulWeightsOnlyGrouped", "g32"), ("QMatMulWeightsOnlyGrouped", "g48")}
    calls = [(n.overload, n.input[1]) for n in out.graph.node if n.domain == "quant"]
    assert [c[0] for c in calls] == ["g32", "g48"]
    out = q_oracle(fixture("block"), CONFIGS["int4_g32"]())               # one size: no overload, the reference's shape
    assert [(f.name, f.overload) for f in out.functions] == [("QMatMulWeightsOnlyGrouped", None)]


def test_quantize_entry_point_routes_bytes_paths_and_parsed_models(tmp_path, monkeypatch):
    import onnx_quantize_amd.model_quantize as MQ
    real = MQ.quantize_model
    monkeypatch.setattr(MQ, "quantize_model", lambda m, qc, **kw: real(m, qc, weight_arrays=oracle_weight_arrays, quantize_bias=O.quantize_bias))
    path = os.path.join(FIXTURES, "mlp_matmul.onnx")
    data = open(path, "rb").read()
    qc = CONFIGS["int8_channel"]()
    out_bytes = quantize(data, qc)
    assert isinstance(out_bytes, bytes) and P.parse_model(out_bytes).functions[0].name == "QMatMulWeightsOnlyQDQ"
    out_msg = quantize(path, qc)
    assert isinstance(out_msg, P.Message) and P.serialize(out_msg) == out_bytes
    assert P.serialize(quantize(P.parse_model(data), qc)) == out_bytes
    assert quantize(data, QConfig()) is data                             # nothing to quantize: the model comes back as it is
    dst = tmp_path / "q.onnx"
    MQ.quantize_file(path, dst, qc, weight_arrays=oracle_weight_arrays, quantize_bias=O.quantize_bias)
    assert dst.read_bytes() == out_bytes
    with pytest.raises(TypeError):
        MQ.as_model(3.5)
    for not_a_model in (3.5, None, ["model.onnx"], {"graph": None}):       # quantize.py:38-41
        with pytest.raises(TypeError, match="model must be"):
            quantize(not_a_model, qc)
    with pytest.raises(TypeError, match="QConfig"):
        real(data, {"weights": None})


def test_calibrated_configurations_need_the_device_by_default():
    """Calibration runs through the HIP library (no CPU fallback): without a GPU the call fails loudly, it does not quantize
    with something else."""
    if torch.cuda.is_available():
        pytest.skip("a GPU is present: tests/test_onnx_model_gpu.py runs these configurations")
    from onnx_quantize_amd.model_quantize import quantize_model
    qc = QConfig(weights=QWeightArgs(dtype=QuantType.QInt8), input_activations=QActivationArgs(dtype=QuantType.QUInt8))
    with pytest.raises(Exception) as e:
        quantize_model(fixture("mlp_matmul"), qc, weight_arrays=oracle_weight_arrays, quantize_bias=O.quantize_bias)
    assert not isinstance(e.value, (AssertionError, KeyError)), e.value


def _act(dt, static=True):
    return QActivationArgs(dtype=QuantType.from_string(dt), is_static=static)


CALIBRATED = {
    # BASELINE config 3: static int8 activations on both sides, QDQ
    "static_qdq": lambda: QConfig(weights=QWeightArgs(dtype=QuantType.QInt8), input_activations=_act("int8"), output_activations=_act("int8")),
    "static_input_only": lambda: QConfig(weights=QWeightArgs(dtype=QuantType.QUInt8, group_size=-1), input_activations=_act("uint8")),
    "static_output_momentum": lambda: QConfig(weights=QWeightArgs(dtype=QuantType.QInt8, symmetric=True), output_activations=_act("uint8"),
                                              calibration_params={"momentum": 0.5, "num_samples": 24, "batch_size": 6}),
    "dynamic_in_out": lambda: QConfig(weights=QWeightArgs(dtype=QuantType.QInt8), input_activations=_act("uint8", False),
                                      output_activations=_act("uint8", False)),
    "qlinear": lambda: QConfig(weights=QWeightArgs(dtype=QuantType.QInt8, symmetric=True), format="qlinear",
                               input_activations=_act("uint8"), output_activations=_act("uint8")),
}


def _feeds(name, gen):
    return {"mlp_gemm": lambda: torch.randn(5, 64, generator=gen), "mlp_matmul": lambda: torch.randn(2, 3, 64, generator=gen),
            "block": lambda: torch.randn(2, 6, 64, generator=gen)}[name]()


@pytest.mark.parametrize("cfg", sorted(CALIBRATED))
@pytest.mark.parametrize("name", ["mlp_gemm", "mlp_matmul", "block"])
def test_calibrated_configurations_with_the_oracle_as_provider(name, cfg):
    """calibrate.py:310-380 + `_get_activation_qparams` (qrules/base.py:15-40): initializers `<node output>/<kind>/scale` and
    `/zero_point` hold what the calibrator computed for the node's input / output value; the call takes them in the order
    of the function's signature."""
    qc = CALIBRATED[cfg]()
    gen = torch.Generator().manual_seed(21)
    data = torch.randn(30, *_feeds(name, gen).shape[1:], generator=gen).numpy()
    qc.calibration_data = data
    src = fixture(name)
    out = P.parse_model(P.serialize(q_oracle(src, qc)))
    inits = {t.name: t for t in out.graph.initializer}
    # the expected ranges, straight from the oracle on the float model's activations
    targets = [n for n in src.graph.node if n.op_type in ("MatMul", "Gemm") and n.input[1] in {t.name for t in src.graph.initializer}]
    params = qc.calibration_params
    batches = O.prepare_calibration_data(data, params.batch_size, params.num_samples)
    names = list(dict.fromkeys([n.input[0] for n in targets] + [n.output[0] for n in targets]))
    runner = GraphRunner(src, outputs=names, device="cpu")
    acts = [{k: v.numpy() for k, v in runner(torch.from_numpy(np.ascontiguousarray(b))).items()} for b in batches]
    cal_in = qc.input_activations is not None and qc.input_activations.is_static
    cal_out = qc.output_activations is not None and qc.output_activations.is_static
    key = lambda a: (a.dtype.key, a.symmetric, a.reduce_range)      # noqa: E731
    flow = O.calibrate_flow([{k: b[k] for k in ([n.input[0] for n in targets] if cal_in else []) + ([n.output[0] for n in targets] if cal_out else [])}
                             for b in acts], [n.input[0] for n in targets] if cal_in else [], [n.output[0] for n in targets] if cal_out else [],
                            params.momentum, key(qc.input_activations) if cal_in else None, key(qc.output_activations) if cal_out else None)
    by_name = {n.name: n for n in out.graph.node}
    for t in targets:
        n = by_name[t.name]
        assert n.domain == "quant"
        fn = next(f for f in out.functions if f.name == n.op_type)
        assert len(n.input) == len(fn.input), (n.op_type, list(n.input), list(fn.input))
        for kind, on, value in (("input", cal_in, t.input[0]), ("output", cal_out, t.output[0])):
            if not on:
                assert f"{t.output[0]}/{kind}/scale" not in inits
                continue
            s, z = flow[(kind, value)]
            prefix = "x" if kind == "input" else "out"
            s_name, z_name = (n.input[list(fn.input).index(f"{prefix}_{p}")] for p in ("scale", "zero_point"))
            # (names may have been folded by DeduplicateInitializersPass: read through the call)
            assert P.tensor_to_numpy(inits[s_name]).tobytes() == np.asarray(s, np.float32).tobytes(), (t.name, kind)
            got_z = P.tensor_to_numpy(inits[z_name])
            assert got_z.dtype == getattr(qc, f"{kind}_activations").dtype.np_dtype and int(got_z) == int(z)
            assert f"{t.output[0]}/{kind}/scale" in inits or s_name != f"{t.output[0]}/{kind}/scale"
    if cfg == "qlinear":
        assert {n.op_type for n in out.graph.node if n.domain} == ({"QLinearMatMul", "QLinearGemm"} if name == "mlp_gemm" else {"QLinearMatMul"})
        if name == "mlp_gemm":
            gemm = by_name["/0/Gemm"]
            assert inits[gemm.input[2]].data_type == P.DataType.INT32            # bias / (x_scale * w_scale) (_qlinear/gemm_to_qgemm.py:48-57)
            assert {o.domain for o in out.opset_import} >= {"quant", "com.microsoft"}
    # the quantized model still computes the model: 8-bit everywhere, a few percent
    feed = _feeds(name, gen)
    want, got = GraphRunner(src, device="cpu")(feed), GraphRunner(out, device="cpu")(feed)
    for k in want:
        rel = ((got[k] - want[k]).norm() / want[k].norm()).item()
        assert rel < (0.25 if "momentum" in cfg else 0.08), (name, cfg, rel)     # an EMA of batch ranges clips the tails


def test_gptq_configuration_with_the_oracle_as_provider():
    """BASELINE config 4's rule path (int4, groups, GPTQ): every node's weight comes from `_gptq` on the concatenated calibration
    input of THAT node (calibrate.py:288-307); q / k / v read the same value and get the same input."""
    from onnx_quantize_amd import GPTQConfig
    qc = QConfig(weights=QWeightArgs(dtype=QuantType.QInt4, group_size=32, algorithm=GPTQConfig(block_size=16)),
                 calibration_params={"num_samples": 16, "batch_size": 4})
    gen = torch.Generator().manual_seed(33)
    data = torch.randn(16, 6, 64, generator=gen).numpy()
    qc.calibration_data = data
    src = fixture("block")
    out = P.parse_model(P.serialize(q_oracle(src, qc)))
    inits = {t.name: t for t in out.graph.initializer}
    runner = GraphRunner(src, outputs=["/ln1/LayerNormalization_output_0"], device="cpu")
    x = np.concatenate([runner(torch.from_numpy(np.ascontiguousarray(b)))["/ln1/LayerNormalization_output_0"].numpy()
                        for b in O.prepare_calibration_data(data, 4, 16)], axis=0)
    for node_name in ("/q/MatMul", "/k/MatMul", "/v/MatMul"):
        n = next(m for m in out.graph.node if m.name == node_name)
        w = P.tensor_to_numpy(next(t for t in src.graph.initializer if t.name == n.input[1]))
        q, s, z = O.seam_arrays(w, "gptq", "int4", "group", 32, False, False, 1.0, False, x=x, nbits=False, block_size=16, percdamp=0.01, actorder=False)
        assert n.op_type == "QMatMulWeightsOnlyGrouped" and inits[n.input[1]].data_type == P.DataType.INT4
        assert np.array_equal(P.tensor_to_numpy(inits[n.input[1]]), q)
        assert P.tensor_to_numpy(inits[n.input[2]]).tobytes() == np.asarray(s).tobytes()
    feed = torch.randn(2, 6, 64, generator=gen)
    want, got = GraphRunner(src, device="cpu")(feed)["y"], GraphRunner(out, device="cpu")(feed)["y"]
    assert ((got - want).norm() / want.norm()).item() < 0.08


# ------------------------------------------------------------------------------------------------------------------ AWQ / SmoothQuant
def _pre_cfg(kind, **weights):
    from onnx_quantize_amd import AwqConfig, SmoothQuantConfig
    pre = {"smooth": lambda: SmoothQuantConfig(alpha=0.5), "awq": lambda: AwqConfig(), "awq_clip": lambda: AwqConfig(clip_search=True)}[kind]()
    return QConfig(weights=QWeightArgs(**weights), preprocessors=[pre], calibration_params={"num_samples": 16, "batch_size": 4})


@pytest.mark.parametrize("kind,weights", [("smooth", dict(dtype=QuantType.QInt8, group_size=-1)),
                                          ("awq", dict(dtype=QuantType.QUInt4, group_size=32)),
                                          ("awq_clip", dict(dtype=QuantType.QInt4, group_size=32))])
def test_preprocessors_rewrite_the_graph_like_the_reference_passes(kind, weights):
    """smooth_quant.py:91-134 / awq.py:114-259: per target node a `Mul` by 1 / scale in front (initializer `<output>_scale`),
    the scale folded into the weight's rows, the shared calibration input divided IN PLACE (so k searches on what q left),
    AWQ's clip ratio per node; then the model is calibrated again (pre_passes/__init__.py:85-88)."""
    gen = torch.Generator().manual_seed(8)
    data = (torch.randn(16, 6, 64, generator=gen) * torch.linspace(0.2, 4.0, 64)).numpy()
    qc = _pre_cfg(kind, **weights)
    qc.calibration_data = data
    src = fixture("block")
    out = P.parse_model(P.serialize(q_oracle(src, qc)))
    inits = {t.name: t for t in out.graph.initializer}
    src_inits = {t.name: P.tensor_to_numpy(t) for t in src.graph.initializer}
    nodes = list(out.graph.node)
    by_name = {n.name: n for n in nodes}
    a = qc.weights
    # the reference's walk restated on the float model: q, k, v share ONE input array, divided in place as the walk proceeds
    taps = {n.name: n.input[0] for n in src.graph.node if n.op_type == "MatMul" and n.input[1] in src_inits}
    runner = GraphRunner(src, outputs=list(dict.fromkeys(taps.values())), device="cpu")
    acts = [{k: v.numpy() for k, v in runner(torch.from_numpy(np.ascontiguousarray(b))).items()} for b in O.prepare_calibration_data(data, 4, 16)]
    shared = O.gptq_inputs(acts)
    for t in [n for n in src.graph.node if n.name in taps]:
        x, w = shared[t.input[0]], src_inits[t.input[1]]
        if kind == "smooth":
            scale = O.smooth_quant_scale(x, w, 0.5)
        else:
            scale, _ = O.awq_scale_search(x, w, a.dtype.key, "group", 32, a.symmetric, a.reduce_range)
        scale = scale.astype(np.float32)
        x /= scale.reshape(1, -1)
        updated = np.multiply(scale.reshape(-1, 1), w)
        call = by_name[t.name]
        mul = nodes[nodes.index(call) - 1]
        assert mul.op_type == "Mul" and list(mul.input) == [t.input[0], f"{t.output[0]}_scale"] and call.input[0] == mul.output[0]
        assert P.tensor_to_numpy(inits[f"{t.output[0]}_scale"]).tobytes() == (1.0 / scale).astype(np.float32).tobytes()
        clip = 1.0
        if kind == "awq_clip":
            clip, _ = O.awq_clip_search(x, updated, a.dtype.key, "group", 32, a.symmetric, a.reduce_range)
        g = O.resolve_group_size(w.shape[0], a.group_size) if a.strategy.value == "group" else a.group_size
        nbits = O.matmul_nbits_compatible(a.dtype.key, a.strategy.value, g)
        q, s, z = O.seam_arrays(updated, "rtn", a.dtype.key, a.strategy.value, g, a.symmetric, a.reduce_range, clip, a.mse, nbits=nbits)
        assert np.array_equal(P.tensor_to_numpy(inits[call.input[1]]), q), t.name
        assert P.tensor_to_numpy(inits[call.input[2]]).tobytes() == np.asarray(s).tobytes(), t.name
    assert len([n for n in nodes if n.op_type == "Mul" and n.name.endswith("/scale_input")]) == 6
    # the rescaled model is the same function; with the quantization error on top it stays close
    feed = torch.from_numpy(data[:3])
    want, got = GraphRunner(src, device="cpu")(feed)["y"], GraphRunner(out, device="cpu")(feed)["y"]
    assert ((got - want).norm() / want.norm()).item() < (0.02 if kind == "smooth" else 0.12)


def test_static_activations_are_recalibrated_after_preprocessing():
    """pre_passes/__init__.py:85-88: the input ranges of a smoothed node are those of the Mul's output."""
    from onnx_quantize_amd import SmoothQuantConfig
    gen = torch.Generator().manual_seed(9)
    data = (torch.randn(12, 3, 64, generator=gen) * torch.linspace(0.2, 4.0, 64)).numpy()
    qc = QConfig(weights=QWeightArgs(dtype=QuantType.QInt8), input_activations=QActivationArgs(dtype=QuantType.QUInt8),
                 preprocessors=[SmoothQuantConfig(alpha=0.6)], calibration_params={"num_samples": 12, "batch_size": 4}, calibration_data=data)
    out = q_oracle(fixture("mlp_matmul"), qc)
    first = next(n for n in out.graph.node if n.domain == "quant")
    mul = next(n for n in out.graph.node if n.op_type == "Mul")
    assert first.input[0] == mul.output[0] and first.op_type == "QMatMulWeightStaticInputQDQ"
    inits = {t.name: P.tensor_to_numpy(t) for t in out.graph.initializer}
    smoothed = data.reshape(-1, 64) * inits[mul.input[1]]
    batches = O.prepare_calibration_data(smoothed.reshape(12, 3, 64), 4, 12)
    cal = O.MinMaxOracle()
    for b in batches:
        cal.collect("v", b)
    s, z = O.qparams(*cal.compute_range("v"), "uint8", False, False)
    assert inits[first.input[4]].tobytes() == np.asarray(s, np.float32).tobytes() and int(inits[first.input[5]]) == int(z)


# ------------------------------------------------------------------------------------------------------------------ contrib operators
def _gqa_model(b, s, heads, kv, d, window=None):
    hid = heads * d
    gen = torch.Generator().manual_seed(0)
    w = {n: torch.randn(hid, width, generator=gen) * 0.1 for n, width in (("wq", hid), ("wk", kv * d), ("wv", kv * d))}
    inv = 1.0 / (10000 ** (torch.arange(0, d, 2).float() / d))
    fr = torch.outer(torch.arange(64).float(), inv)
    attrs = dict(num_heads=heads, kv_num_heads=kv, do_rotary=1)
    if window:
        attrs["local_window_size"] = window
    nodes = [P.make_node("MatMul", ["x", "wq"], ["q"]), P.make_node("MatMul", ["x", "wk"], ["k"]), P.make_node("MatMul", ["x", "wv"], ["v"]),
             P.make_node("GroupQueryAttention", ["q", "k", "v", "pk", "pv", "seqlens", "total", "cos", "sin"], ["y", "prk", "prv"],
                         domain="com.microsoft", **attrs)]
    inits = [P.numpy_to_tensor(n, a.numpy()) for n, a in list(w.items()) + [("cos", fr.cos()), ("sin", fr.sin())]]
    g = P.Message("GraphProto", name="g", node=nodes, initializer=inits, output=[P.make_value_info("y", 1, None)],
                  input=[P.make_value_info("x", 1, [b, s, hid]), P.make_value_info("pk", 1, [b, kv, "p", d]), P.make_value_info("pv", 1, [b, kv, "p", d]),
                         P.make_value_info("seqlens", 6, [b]), P.make_value_info("total", 6, [])])
    model = P.Message("ModelProto", ir_version=10, graph=g, opset_import=[P.Message("OperatorSetIdProto", domain="", version=21),
                                                                           P.Message("OperatorSetIdProto", domain="com.microsoft", version=1)])
    return model, w, fr.cos(), fr.sin()


def test_contrib_operators_of_genai_style_exports():
    """`com.microsoft::GroupQueryAttention` (rotary inside, grouped heads, causal + sliding window + right padding, prefill
    and decode), `RotaryEmbedding`, the RMS-norm family and the Gelu variants against plain torch restatements."""
    b, s, heads, kv, d = 2, 7, 4, 2, 16
    hid = heads * d
    model, w, cos, sin = _gqa_model(b, s, heads, kv, d)
    x = torch.randn(b, s, hid, generator=torch.Generator().manual_seed(1))
    empty = torch.zeros(b, kv, 0, d)

    def feed(xs, pk=empty, pv=empty, valid=None):
        total = pk.shape[2] + xs.shape[1]
        lens = torch.full((b,), total - 1, dtype=torch.int32) if valid is None else torch.tensor(valid, dtype=torch.int32) - 1
        return {"x": xs, "pk": pk, "pv": pv, "seqlens": lens, "total": torch.tensor(total, dtype=torch.int32)}

    def rope(t, n):                                                        # HF `rotate_half` convention = non-interleaved
        c, s_ = torch.cat((cos[:n], cos[:n]), -1)[None, None], torch.cat((sin[:n], sin[:n]), -1)[None, None]
        return t * c + torch.cat((-t[..., d // 2:], t[..., : d // 2]), -1) * s_

    def reference(xs, mask):
        n = xs.shape[1]
        q = rope((xs @ w["wq"]).view(b, n, heads, d).transpose(1, 2), n)
        k = rope((xs @ w["wk"]).view(b, n, kv, d).transpose(1, 2), n).repeat_interleave(heads // kv, 1)
        v = (xs @ w["wv"]).view(b, n, kv, d).transpose(1, 2).repeat_interleave(heads // kv, 1)
        return torch.nn.functional.scaled_dot_product_attention(q, k, v, attn_mask=mask).transpose(1, 2).reshape(b, n, hid)

    causal = torch.ones(s, s, dtype=torch.bool).tril()
    run = GraphRunner(model, outputs=["y", "prk", "prv"], device="cpu")
    full = run(feed(x))
    torch.testing.assert_close(full["y"], reference(x, causal), rtol=1e-5, atol=1e-5)
    assert full["prk"].shape == (b, kv, s, d)
    # decode: the last token against the cache of the first s - 1 equals the last row of the prefill
    pre = run(feed(x[:, : s - 1]))
    dec = run(feed(x[:, s - 1:], pre["prk"], pre["prv"]))
    torch.testing.assert_close(dec["y"][:, 0], full["y"][:, -1], rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(dec["prk"], full["prk"], rtol=1e-5, atol=1e-5)
    # sliding window: the token itself and `window` tokens to its left
    wmodel, *_ = _gqa_model(b, s, heads, kv, d, window=2)
    idx = torch.arange(s)
    band = causal & (idx[None, :] >= idx[:, None] - 2)
    torch.testing.assert_close(GraphRunner(wmodel, device="cpu")(feed(x))["y"], reference(x, band), rtol=1e-5, atol=1e-5)
    # right padding: row 1 has 4 valid tokens; its first 4 outputs are those of the 4-token prompt
    padded = run(feed(x, valid=[s, 4]))
    short = run(feed(x[:, :4]))
    torch.testing.assert_close(padded["y"][1, :4], short["y"][1], rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(padded["y"][0], full["y"][0], rtol=1e-5, atol=1e-5)

    def one(op, inputs, outputs=("y",), domain="com.microsoft", **attrs):
        names = ["" if v is None else f"i{k}" for k, v in enumerate(inputs)]           # None: an optional input left out
        given = [(n, v) for n, v in zip(names, inputs) if v is not None]
        g = P.Message("GraphProto", name="g", node=[P.make_node(op, names, list(outputs), domain=domain, **attrs)],
                      input=_declared(*zip(*given)), output=[P.make_value_info(o, 1, None) for o in outputs if o])
        m = P.Message("ModelProto", ir_version=10, graph=g, opset_import=[P.Message("OperatorSetIdProto", domain="", version=21)])
        return GraphRunner(m, device="cpu")(dict(given))

    # MultiHeadAttention: separate q / k / v with a bias, an additive attention bias, a past, causal
    mh, md = 4, 8
    mq, mk, mv, mb = torch.randn(2, 3, 32), torch.randn(2, 3, 32), torch.randn(2, 3, 32), torch.randn(96)
    pk, pv, ab = torch.randn(2, mh, 5, md), torch.randn(2, mh, 5, md), torch.randn(1, mh, 3, 8)
    got = one("MultiHeadAttention", [mq, mk, mv, mb, None, ab, pk, pv], outputs=("y", "pk", "pv"), num_heads=mh, unidirectional=1)
    split = lambda t, o: (t + mb[o:o + 32]).reshape(2, 3, mh, md).transpose(1, 2)          # noqa: E731
    kk, vv = torch.cat((pk, split(mk, 32)), 2), torch.cat((pv, split(mv, 64)), 2)
    sc = split(mq, 0) @ kk.transpose(-1, -2) / md ** 0.5 + ab
    sc = sc.masked_fill(torch.arange(8).reshape(1, -1) > (5 + torch.arange(3)).reshape(-1, 1), float("-inf"))
    torch.testing.assert_close(got["y"], (torch.softmax(sc, -1) @ vv).transpose(1, 2).reshape(2, 3, 32), rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(got["pk"], kk)
    torch.testing.assert_close(got["pv"], vv)

    t, skip, gamma = torch.randn(2, 5, 32), torch.randn(2, 5, 32), torch.rand(32) + 0.5
    rms = lambda v: v * torch.rsqrt(v.pow(2).mean(-1, keepdim=True) + 1e-6) * gamma      # noqa: E731
    torch.testing.assert_close(one("SimplifiedLayerNormalization", [t, gamma], domain="", epsilon=1e-6, axis=-1)["y"], rms(t))
    got = one("SkipSimplifiedLayerNormalization", [t, skip, gamma], outputs=("y", "", "", "sum"), epsilon=1e-6)
    torch.testing.assert_close(got["y"], rms(t + skip))
    torch.testing.assert_close(got["sum"], t + skip)
    beta = torch.randn(32)
    got = one("SkipLayerNormalization", [t, skip, gamma, beta], outputs=("y", "", "", "sum"), epsilon=1e-5)
    torch.testing.assert_close(got["y"], torch.nn.functional.layer_norm(t + skip, (32,), gamma, beta, 1e-5))
    torch.testing.assert_close(one("FastGelu", [t, beta])["y"], torch.nn.functional.gelu(t + beta, approximate="tanh"))
    torch.testing.assert_close(one("BiasGelu", [t, beta])["y"], torch.nn.functional.gelu(t + beta))
    torch.testing.assert_close(one("QuickGelu", [t], alpha=1.702)["y"], t * torch.sigmoid(1.702 * t))
    # RotaryEmbedding on [B, S, H * D] with explicit positions, both layouts
    pos = torch.arange(3, 3 + s).reshape(1, s).expand(b, s).contiguous()
    xq = (x @ w["wq"])
    for inter in (0, 1):
        got = one("RotaryEmbedding", [xq, pos, cos, sin], interleaved=inter, num_heads=heads)["y"].view(b, s, heads, d).transpose(1, 2)
        th = xq.view(b, s, heads, d).transpose(1, 2)
        c, s_ = cos[pos][:, None], sin[pos][:, None]
        if inter:
            x1, x2 = th[..., 0::2], th[..., 1::2]
            want = torch.stack((x1 * c - x2 * s_, x2 * c + x1 * s_), -1).flatten(-2)
        else:
            x1, x2 = th[..., : d // 2], th[..., d // 2:]
            want = torch.cat((x1 * c - x2 * s_, x2 * c + x1 * s_), -1)
        torch.testing.assert_close(got, want, rtol=1e-5, atol=1e-5)


def test_graph_runner_gets_through_a_convolutional_front_end():
    """A CNN whose classifier is a Gemm: Conv (groups, strides, padding, bias), BatchNormalization, MaxPool / AveragePool /
    GlobalAveragePool, HardSwish, Flatten -- exported by torch, run against the module, quantized (the classifier only)."""
    import io
    import warnings
    from torch.onnx._internal.torchscript_exporter import onnx_proto_utils
    onnx_proto_utils._add_onnxscript_fn = lambda proto, _ops: proto
    torch.manual_seed(0)
    net = torch.nn.Sequential(
        torch.nn.Conv2d(3, 8, 3, stride=2, padding=1), torch.nn.BatchNorm2d(8), torch.nn.ReLU(), torch.nn.MaxPool2d(2),
        torch.nn.Conv2d(8, 16, 3, padding=1, groups=4, bias=False), torch.nn.Hardswish(), torch.nn.AvgPool2d(2, padding=1),
        torch.nn.AdaptiveAvgPool2d(1), torch.nn.Flatten(), torch.nn.Linear(16, 10)).eval()
    with torch.no_grad():
        net[1].running_mean.normal_()
        net[1].running_var.uniform_(0.5, 2.0)
    f = io.BytesIO()
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        torch.onnx.export(net, (torch.randn(2, 3, 32, 32),), f, dynamo=False, opset_version=17, input_names=["image"], output_names=["scores"],
                          dynamic_axes={"image": {0: "batch"}, "scores": {0: "batch"}}, do_constant_folding=False)
    data = f.getvalue()
    model = P.parse_model(data)
    assert P.serialize(model) == data
    assert {"Conv", "MaxPool", "AveragePool", "GlobalAveragePool", "HardSwish", "Gemm"} <= {n.op_type for n in model.graph.node}
    x = torch.randn(5, 3, 32, 32)
    with torch.no_grad():
        torch.testing.assert_close(GraphRunner(model, device="cpu")(x)["scores"], net(x), rtol=1e-4, atol=1e-5)
    out = q_oracle(model, CONFIGS["int8_channel"]())
    assert [n.op_type for n in out.graph.node if n.domain] == ["QGemmWeightsOnlyQDQ"]
    assert sum(n.op_type == "Conv" for n in out.graph.node) == 2
    got, want = GraphRunner(out, device="cpu")(x)["scores"], GraphRunner(model, device="cpu")(x)["scores"]
    assert ((got - want).norm() / want.norm()).item() < 0.02


@pytest.mark.parametrize("opset", [13, 17])
def test_pad_and_resize_exports_run_and_cross_the_opset_change(opset):
    """Pad and Resize only gained optional inputs / attributes between opset 13 and 21: such nodes pass `_raise_opset` as they are,
    and the runner computes what torch computes for the forms torch's exporter writes."""
    import io
    import warnings
    from torch.onnx._internal.torchscript_exporter import onnx_proto_utils
    onnx_proto_utils._add_onnxscript_fn = lambda proto, _ops: proto
    F = torch.nn.functional
    pads = {"reflect": lambda x: F.pad(x, (1, 2, 2, 1), mode="reflect"), "constant": lambda x: F.pad(x, (1, 2, 0, 1), value=0.5),
            "replicate": lambda x: F.pad(x, (2, 0, 1, 1), mode="replicate"), "crop": lambda x: F.pad(x, (-1, 2, 1, -2))}
    resizes = {"nearest": lambda x: F.interpolate(x, scale_factor=2.0, mode="nearest"),     # (a scale like 1.7 is stored as float32: other pixels at exact multiples)
               "bilinear": lambda x: F.interpolate(x, scale_factor=1.5, mode="bilinear", align_corners=False),
               "corners": lambda x: F.interpolate(x, size=(9, 11), mode="bilinear", align_corners=True),
               "bicubic": lambda x: F.interpolate(x, scale_factor=2.0, mode="bicubic", align_corners=False)}

    class Net(torch.nn.Module):
        def __init__(self, pad, resize):
            super().__init__()
            self.pad, self.resize, self.conv, self.fc = pad, resize, torch.nn.Conv2d(3, 4, 3), torch.nn.Linear(4, 16)

        def forward(self, x):
            return self.fc(self.resize(self.conv(self.pad(x))).mean((2, 3)))

    for (pname, pad), (rname, resize) in zip(pads.items(), resizes.items()):
        torch.manual_seed(0)
        net = Net(pad, resize).eval()
        f = io.BytesIO()
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            torch.onnx.export(net, (torch.randn(2, 3, 8, 10),), f, dynamo=False, opset_version=opset, input_names=["x"], output_names=["y"],
                              dynamic_axes={"x": {0: "b"}})
        model = P.parse_model(f.getvalue())
        assert {"Pad", "Resize"} <= {n.op_type for n in model.graph.node}, (pname, rname)
        x = torch.randn(3, 3, 8, 10)
        with torch.no_grad():
            torch.testing.assert_close(GraphRunner(model, device="cpu")(x)["y"], net(x), rtol=1e-5, atol=1e-6)
        out = q_oracle(model, CONFIGS["int8_channel"]())
        assert [n.op_type for n in out.graph.node if n.domain] == ["QGemmWeightsOnlyQDQ"] and out.opset_import[0].version == 21
        got, want = GraphRunner(out, device="cpu")(x)["y"], GraphRunner(model, device="cpu")(x)["y"]
        assert ((got - want).norm() / want.norm()).item() < 0.02
    # forms the runner does not compute are refused by name
    with pytest.raises(UnsupportedOperator, match="Resize"):
        _run_one("Resize", [np.zeros((1, 1, 4, 4), np.float32), np.zeros(0, np.float32), np.array([1, 1, 2, 2], np.float32)], mode="nearest",
                 coordinate_transformation_mode="half_pixel")


def test_more_operators_against_torch_exports_and_their_definitions():
    """Operators around MatMul / Gemm heads that a calibration walk has to get through: each computed by the runner from torch's
    own export of the torch function, or -- where the exporter writes other operators for it -- from the operator's definition."""
    import io
    import warnings
    from torch.onnx._internal.torchscript_exporter import onnx_proto_utils
    onnx_proto_utils._add_onnxscript_fn = lambda proto, _ops: proto
    F = torch.nn.functional
    g = torch.Generator().manual_seed(0)
    r = lambda *sh: torch.randn(*sh, generator=g)                             # noqa: E731

    class M(torch.nn.Module):
        def __init__(self, fn, mods=()):
            super().__init__()
            self.fn, self.mods = fn, torch.nn.ModuleList(mods)

        def forward(self, *xs):
            return self.fn(self.mods, *xs)

    cases = {
        "ReduceProd": (lambda m, x: x.prod(dim=1), (r(3, 4, 5),)),
        "ReduceL2": (lambda m, x: x.norm(dim=(1, 2)), (r(3, 4, 5),)),
        "ReduceL1": (lambda m, x: x.norm(p=1, dim=2), (r(3, 4, 5),)),
        "ReduceLogSumExp": (lambda m, x: x.logsumexp(dim=2), (r(3, 4, 5),)),
        "ArgMin": (lambda m, x: x.argmin(dim=1), (r(3, 4, 5),)),
        "TopK": (lambda m, x: torch.topk(x, 3, dim=2)[0] + torch.topk(x, 3, dim=2)[1], (r(3, 4, 5),)),
        "NonZero": (lambda m, x: (x > 0).nonzero(), (r(3, 4),)),
        "OneHot": (lambda m, x: F.one_hot(x, 7).float(), (torch.randint(0, 7, (3, 4), generator=g),)),
        "ScatterElements": (lambda m, x, i, u: x.scatter(1, i, u), (r(3, 6), torch.randint(0, 6, (3, 2), generator=g), r(3, 2))),
        "ScatterND": (lambda m, x, u: x.index_put((torch.tensor([0, 2]), torch.tensor([1, 3])), u), (r(3, 4, 5), r(2, 5))),
        "DepthToSpace": (lambda m, x: F.pixel_shuffle(x, 2), (r(2, 8, 3, 3),)),
        "InstanceNormalization": (lambda m, x: m[0](x), (r(2, 4, 5, 5),), [torch.nn.InstanceNorm2d(4, affine=True)]),
        "ConvTranspose": (lambda m, x: m[0](x) + m[1](x[:, :, 0]).sum(), (r(2, 4, 5, 5),),
                          [torch.nn.ConvTranspose2d(4, 6, 3, stride=2, padding=1, output_padding=1), torch.nn.ConvTranspose1d(4, 4, 4, stride=2, groups=2, bias=False)]),
        "Selu": (lambda m, x: F.selu(x), (r(3, 4),)),
        "Celu": (lambda m, x: F.celu(x, 1.3), (r(3, 4),)),
        "Atan": (lambda m, x: x.atan() + x.tan() + (x * 0.5).asin() + (x * 0.5).acos(), (r(3, 4).clamp(-1, 1),)),
        "Xor": (lambda m, x, y: (x > 0) ^ (y > 0), (r(3, 4), r(3, 4))),
    }
    for name, case in cases.items():
        fn, args = case[:2]
        net = M(fn, case[2] if len(case) > 2 else ()).eval()
        f = io.BytesIO()
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            torch.onnx.export(net, args, f, dynamo=False, opset_version=17, input_names=[f"x{i}" for i in range(len(args))], output_names=["y"])
        model = P.parse_model(f.getvalue())
        assert name in {n.op_type for n in model.graph.node}, name
        got = GraphRunner(model, device="cpu")({f"x{i}": a for i, a in enumerate(args)})["y"]
        with torch.no_grad():
            want = net(*args)
        assert got.shape == want.shape and got.dtype == want.dtype, name
        torch.testing.assert_close(got, want, rtol=1e-5, atol=1e-6, msg=name)

    # by definition (onnx/defs): operators the exporter writes differently
    x = r(2, 8, 4, 6).numpy()
    b, c, h, w = x.shape
    np.testing.assert_array_equal(_run_one("SpaceToDepth", [x], blocksize=2).numpy(),
                                  x.reshape(b, c, h // 2, 2, w // 2, 2).transpose(0, 3, 5, 1, 2, 4).reshape(b, c * 4, h // 2, w // 2))
    back = _run_one("DepthToSpace", [_run_one("SpaceToDepth", [x], blocksize=2).numpy()], blocksize=2, mode="DCR")
    np.testing.assert_array_equal(back.numpy(), x)
    crd = x.reshape(b, c // 4, 2, 2, h, w).transpose(0, 1, 4, 2, 5, 3).reshape(b, c // 4, h * 2, w * 2)
    np.testing.assert_array_equal(_run_one("DepthToSpace", [x], blocksize=2, mode="CRD").numpy(), crd)
    v = r(3, 5).numpy()
    np.testing.assert_allclose(_run_one("Mish", [v]).numpy(), v * np.tanh(np.log1p(np.exp(v))), rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(_run_one("Softsign", [v]).numpy(), v / (1 + np.abs(v)), rtol=1e-6)
    np.testing.assert_array_equal(_run_one("ThresholdedRelu", [v], alpha=0.3).numpy(), np.where(v > 0.3, v, 0).astype(np.float32))
    np.testing.assert_allclose(_run_one("Shrink", [v], lambd=0.4, bias=0.1).numpy(), np.where(v < -0.4, v + 0.1, np.where(v > 0.4, v - 0.1, 0)), rtol=1e-6)
    np.testing.assert_allclose(_run_one("ReduceSumSquare", [v, np.array([1])], keepdims=0).numpy(), (v * v).sum(1), rtol=1e-5)
    np.testing.assert_allclose(_run_one("ReduceLogSum", [np.abs(v), np.array([0])], keepdims=1).numpy(), np.log(np.abs(v).sum(0, keepdims=True)), rtol=1e-5)
    np.testing.assert_allclose(_run_one("Sinh", [v]).numpy() + _run_one("Cosh", [v]).numpy(), np.exp(v), rtol=1e-5)
    np.testing.assert_allclose(_run_one("Mean", [v, 2 * v, 3 * v]).numpy(), 2 * v, rtol=1e-6)
    np.testing.assert_allclose(_run_one("LpNormalization", [v], axis=1, p=2).numpy(), v / np.sqrt((v * v).sum(1, keepdims=True)), rtol=1e-5)
    np.testing.assert_allclose(_run_one("LpNormalization", [v], axis=0, p=1).numpy(), v / np.abs(v).sum(0, keepdims=True), rtol=1e-5)
    fa, fb = r(2, 5, 3).numpy(), r(2, 5, 4).numpy()
    np.testing.assert_allclose(_run_one("FusedMatMul", [fa, fb], transA=1, alpha=0.5).numpy(), 0.5 * np.matmul(fa.transpose(0, 2, 1), fb), rtol=1e-5, atol=1e-6)
    data = r(4, 5, 6).numpy()
    idx = np.array([[0, 1], [3, -1], [-2, 2]], dtype=np.int64)
    np.testing.assert_array_equal(_run_one("GatherND", [data, idx]).numpy(), np.stack([data[0, 1], data[3, 4], data[2, 2]]))
    upd = r(3, 6).numpy()
    want = data.copy()
    want[0, 1], want[3, 4], want[2, 2] = upd
    np.testing.assert_array_equal(_run_one("ScatterND", [data, idx, upd]).numpy(), want)
    added = data.copy()
    for (i, j), u in zip(idx, upd):
        added[i, j] += u
    np.testing.assert_allclose(_run_one("ScatterND", [data, idx, upd], reduction="add").numpy(), added, rtol=1e-6)
    gn = r(2, 6, 3, 3).numpy()
    scale, bias = r(6).numpy(), r(6).numpy()
    grouped = gn.reshape(2, 3, -1)
    normed = ((grouped - grouped.mean(-1, keepdims=True)) / np.sqrt(grouped.var(-1, keepdims=True) + 1e-5)).reshape(gn.shape)
    np.testing.assert_allclose(_run_one("GroupNormalization", [gn, scale, bias], num_groups=3).numpy(),
                               normed * scale[None, :, None, None] + bias[None, :, None, None], rtol=1e-4, atol=1e-5)
    hot = _run_one("OneHot", [np.array([[1, -1], [5, 2]], dtype=np.int64), np.array([4], dtype=np.int64), np.array([-1.0, 2.0], dtype=np.float32)], axis=1)
    want_hot = np.full((2, 4, 2), -1.0, np.float32)
    want_hot[0, 1, 0] = want_hot[0, 3, 1] = want_hot[1, 2, 1] = 2.0                  # index 5 is out of range: its row stays "off"
    np.testing.assert_array_equal(hot.numpy(), want_hot)


def test_a_shared_weight_that_cannot_be_duplicated_is_refused_at_its_second_consumer():
    w = np.random.default_rng(0).standard_normal((8, 8)).astype(np.float32)
    g = P.Message("GraphProto", name="g", input=[P.make_value_info("x", 1, ["n", 8])], output=[P.make_value_info("y", 1, None), P.make_value_info("w", 1, [8, 8])],
                  node=[P.make_node("MatMul", ["x", "w"], ["h"], name="a"), P.make_node("MatMul", ["h", "w"], ["y"], name="b")], initializer=[P.numpy_to_tensor("w", w)])
    model = P.Message("ModelProto", ir_version=10, graph=g, opset_import=[P.Message("OperatorSetIdProto", domain="", version=21)])
    with pytest.raises(ValueError, match="could not be duplicated"):
        q_oracle(model, CONFIGS["int8_channel"]())
    with pytest.raises(ValueError, match="could not be duplicated"):          # the other reader is not a target: it would read the integers
        q_oracle(model, QConfig(weights=QWeightArgs(dtype=QuantType.QInt8), ignore=["^b$"]))
    g.output = [g.output[0]]                                                  # no longer an output: duplicated, both quantized
    out = q_oracle(model, CONFIGS["int8_channel"]())
    assert [n.op_type for n in out.graph.node] == ["QMatMulWeightsOnlyQDQ"] * 2


def test_a_model_can_be_written_over_the_files_its_tensors_are_mapped_from(tmp_path):
    """Both files are written under temporary names and renamed: quantizing a file onto itself does not truncate the side file under
    the live memory maps, and no partial file is ever visible under the final names."""
    import onnx_quantize_amd.model_quantize as MQ
    src = tmp_path / "m.onnx"
    P.save_model(fixture("block"), src, external_data="m.onnx.data", size_threshold=16)
    loaded = P.load_model(src)                                                   # tensors mapped from m.onnx.data
    before = {t.name: P.tensor_to_numpy(t).copy() for t in loaded.graph.initializer}
    P.save_model(loaded, src, external_data="m.onnx.data", size_threshold=16)    # onto itself
    again = P.load_model(src)
    assert all(np.array_equal(before[t.name], P.tensor_to_numpy(t)) for t in again.graph.initializer)
    assert all(np.array_equal(before[t.name], P.tensor_to_numpy(t)) for t in loaded.graph.initializer)      # the old maps read the old inode
    want = P.serialize(q_oracle(fixture("block"), CONFIGS["int8_channel"]()))
    MQ.quantize_file(src, src, CONFIGS["int8_channel"](), external_data="m.onnx.data", weight_arrays=oracle_weight_arrays, quantize_bias=O.quantize_bias)
    out = P.load_model(src)
    assert sum(bool(n.domain) for n in out.graph.node) > 0
    got = {t.name: P.tensor_to_numpy(t) for t in out.graph.initializer}
    ref = {t.name: P.tensor_to_numpy(t) for t in P.parse_model(want).graph.initializer}
    assert set(got) == set(ref) and all(np.array_equal(got[k], ref[k]) for k in ref)
    assert sorted(os.listdir(tmp_path)) == ["m.onnx", "m.onnx.data"]             # no temporary file left behind


def test_feeds_are_checked_like_a_session_checks_them():
    """calibrate.py:204-251 hands the user's arrays to an onnxruntime session, which refuses unknown names and other element types;
    so does the runner (a float feed used as token ids would otherwise run to nonsense)."""
    runner = GraphRunner(fixture("tied"), device="cpu")
    assert runner.input_names == ["ids"]
    ids = np.zeros((2, 7), np.int64)
    assert runner(ids)["logits"].shape == (2, 7, 32)
    with pytest.raises(TypeError, match="declared int64"):
        runner(ids.astype(np.float32))
    with pytest.raises(TypeError, match="declared int64"):
        runner({"ids": torch.zeros(2, 7, dtype=torch.int32)})
    with pytest.raises(ValueError, match="not an input of the model"):
        runner({"ids": ids, "mask": ids})
    with pytest.raises(KeyError, match="no data for model input"):
        runner({})


def test_bench_and_example_models_are_well_formed_and_quantize():
    """The synthetic models of bench_model.py and examples/gemma3_shapes/gemma3_onnx_file.py (toy sizes): structurally sound, run in
    the graph runner, and go through the writer (oracle providers) with `lm_head` ignored like in the reference's examples."""
    import sys
    sys.path.insert(0, ROOT)
    import bench_model
    model = bench_model.build_model(2, 64, 128)
    P.check_model(model)
    x = torch.randn(2, 5, 64)
    out = q_oracle(model, CONFIGS["uint4_g32"]())
    assert sum(n.op_type == "MatMulNBits" for n in out.graph.node) == 14
    assert ((GraphRunner(out, device="cpu")(x)["y"] - GraphRunner(model, device="cpu")(x)["y"]).norm()).item() > 0
    spec = importlib.util.spec_from_file_location("gemma3_onnx_file", os.path.join(ROOT, "examples", "gemma3_shapes", "gemma3_onnx_file.py"))
    ex = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ex)
    g = ex.build_model(layers=2, vocab=128, max_positions=64)
    P.check_model(g)
    feed = {k: torch.from_numpy(v) for k, v in ex.make_calibration_data(2, 128, 2, 9).items()}
    want = GraphRunner(g, outputs=["logits", "present.1.key"], device="cpu")(feed)
    assert want["logits"].shape == (2, 9, 128) and want["present.1.key"].shape == (2, 1, 9, 256)
    q = q_oracle(g, QConfig(weights=QWeightArgs(dtype="int8", strategy="group", group_size=128), ignore=["lm_head"]))
    assert sum(n.domain == "quant" for n in q.graph.node) == 14 and any(n.name == "/lm_head/MatMul" and not n.domain for n in q.graph.node)
    got = GraphRunner(q, outputs=["logits"], device="cpu")(feed)["logits"]
    assert ((got - want["logits"]).norm() / want["logits"].norm()).item() < 0.05


def _run_one(op, inputs, n_out=1, **attrs):
    names = [f"i{k}" for k in range(len(inputs))]
    outs = [f"o{k}" for k in range(n_out)]
    g = P.Message("GraphProto", name="g", node=[P.make_node(op, names, outs, **attrs)],
                  input=_declared(names, inputs), output=[P.make_value_info(o, 1, None) for o in outs])
    m = P.Message("ModelProto", ir_version=10, graph=g, opset_import=[P.Message("OperatorSetIdProto", domain="", version=21)])
    got = GraphRunner(P.parse_model(P.serialize(m)), device="cpu")(dict(zip(names, inputs)))
    return [got[o] for o in outs] if n_out > 1 else got["o0"]


def test_single_operators_of_the_runner_against_torch():
    """The operators no exported fixture happens to contain, one node at a time against their torch / ONNX-spec meaning."""
    gen = torch.Generator().manual_seed(0)
    x = torch.randn(3, 4, 5, generator=gen)
    i64 = lambda *v: torch.tensor(v, dtype=torch.int64)                       # noqa: E731
    torch.testing.assert_close(_run_one("CumSum", [x, i64(1).reshape(())]), x.cumsum(1))
    torch.testing.assert_close(_run_one("CumSum", [x, i64(2).reshape(())], exclusive=1, reverse=1), x.flip(2).cumsum(2).flip(2) - x)
    assert torch.equal(_run_one("ArgMax", [x], axis=2, keepdims=0), x.argmax(2))
    idx = torch.randint(-4, 4, (3, 2, 5), generator=gen)
    assert torch.equal(_run_one("GatherElements", [x, idx], axis=1), torch.gather(x, 1, idx % 4))
    assert torch.equal(_run_one("Mod", [i64(-7, 7, 5), i64(3, -3, 3)]), i64(2, -2, 2))                 # the divisor's sign
    assert torch.equal(_run_one("Mod", [i64(-7, 7, 5), i64(3, -3, 3)], fmod=1), i64(-1, 1, 2))        # the dividend's sign
    assert torch.equal(_run_one("Div", [i64(-7, 7), i64(2, -2)]), i64(-3, -3))                         # integers truncate towards zero
    torch.testing.assert_close(_run_one("HardSigmoid", [x], alpha=0.25, beta=0.4), (x * 0.25 + 0.4).clamp(0, 1))
    torch.testing.assert_close(_run_one("Elu", [x], alpha=1.5), torch.nn.functional.elu(x, 1.5))
    torch.testing.assert_close(_run_one("PRelu", [x, torch.tensor(0.2)]), torch.where(x >= 0, x, 0.2 * x))
    torch.testing.assert_close(_run_one("Einsum", [x, x.transpose(1, 2).contiguous()], equation="bij,bjk->bik"), x @ x.transpose(1, 2))
    a, b = _run_one("Split", [x, i64(1, 3)], n_out=2, axis=1)
    assert a.shape == (3, 1, 5) and torch.equal(b, x[:, 1:])
    a, b, c = _run_one("Split", [torch.arange(7.0)], n_out=3, axis=0, num_outputs=3)                   # uneven: the last chunk is shorter
    assert [t.numel() for t in (a, b, c)] == [3, 3, 1]
    torch.testing.assert_close(_run_one("Slice", [x, i64(3), i64(-5), i64(2), i64(-1)]), x[:, :, [3, 2, 1]])          # a negative step
    torch.testing.assert_close(_run_one("Slice", [x, i64(1, 0), i64(3, 4), i64(0, 2), i64(1, 2)]), x[1:3, :, 0:4:2])
    assert _run_one("Expand", [torch.ones(4, 1), i64(3, 1, 5)]).shape == (3, 4, 5)
    assert torch.equal(_run_one("Trilu", [x, i64(1).reshape(())], upper=1), x.triu(1))
    torch.testing.assert_close(_run_one("ReduceMean", [x, i64(0, 2)], keepdims=0), x.mean((0, 2)))
    torch.testing.assert_close(_run_one("ReduceSum", [x]), x.sum().reshape(1, 1, 1))
    assert torch.equal(_run_one("ReduceMax", [x, torch.zeros(0, dtype=torch.int64)], noop_with_empty_axes=1), x)
    torch.testing.assert_close(_run_one("Clip", [x, torch.tensor(-0.5), torch.tensor(0.25)]), x.clamp(-0.5, 0.25))
    assert torch.equal(_run_one("Where", [x > 0, x, torch.zeros(())]), torch.where(x > 0, x, torch.zeros(())))
    assert torch.equal(_run_one("Unsqueeze", [x, i64(0, -1)]), x[None, ..., None])
    assert _run_one("Squeeze", [torch.ones(1, 3, 1), i64(0)]).shape == (3, 1)
    assert torch.equal(_run_one("ConstantOfShape", [i64(2, 3)], value=P.numpy_to_tensor("", np.array([7], dtype=np.int64))), torch.full((2, 3), 7))
    assert torch.equal(_run_one("Range", [i64(2).reshape(()), i64(11).reshape(()), i64(3).reshape(())]), torch.arange(2, 11, 3))
    assert torch.equal(_run_one("Shape", [x], start=1), i64(4, 5)) and int(_run_one("Size", [x])) == 60
    torch.testing.assert_close(_run_one("Gemm", [x[0], x[1], torch.ones(4)], transB=1, alpha=0.5, beta=2.0), 0.5 * x[0] @ x[1].T + 2.0)
    q = torch.randint(0, 255, (6, 8), generator=gen).to(torch.uint8)
    s, z = torch.rand(6, 2, generator=gen) + 0.1, torch.randint(0, 255, (6, 2), generator=gen).to(torch.uint8)
    want = (q.float().reshape(6, 2, 4) - z.float()[:, :, None]) * s[:, :, None]
    torch.testing.assert_close(_run_one("DequantizeLinear", [q, s, z], axis=1, block_size=4), want.reshape(6, 8))     # blocked along the last axis
    ch = (q.float() - z[:, 0].float()[:, None]) * s[:, 0][:, None]
    torch.testing.assert_close(_run_one("DequantizeLinear", [q, s[:, 0].contiguous(), z[:, 0].contiguous()], axis=0), ch)
    got = _run_one("QuantizeLinear", [torch.tensor([0.5, 1.5, 2.5, -300.0, 300.0]), torch.tensor(1.0), torch.tensor(0, dtype=torch.int8)])
    assert got.tolist() == [0, 2, 2, -128, 127]                                # half to even, saturated
