"""ONNX model files without the `onnx` package: a protobuf wire codec for the messages of `onnx.proto` a quantized model is
made of (the emitted graph).

The reference hands `onnx.ModelProto` / `onnx_ir.Model` objects around (quantize.py:28-80) and serialises through
`ir.to_proto`; this image has neither package, and a GPU box that only quantizes should not need them.  The wire format
is small: varints, 32 / 64-bit scalars and length-delimited records, with the field numbers of the published `onnx.proto`
(IR version 10 / 11 schema, proto2 syntax).  `parse_model(bytes)` gives a tree of `Message` objects with attribute access
(`model.graph.node[0].op_type`), `serialize(message)` gives the bytes back: fields in field-number order, repeated
scalars packed exactly where `onnx.proto` says `[packed = true]`, absent optional fields absent, fields this table does
not know kept verbatim -- a parse / serialize round trip of a file another producer wrote is byte-identical
(tests/test_onnx_model.py checks that on files written by torch's C++ exporter).  Tensors kept in side files (`external_data`,
what every model past protobuf's 2 GiB limit uses) are memory-mapped on load and can be written back the same way.

Tensors: `tensor_to_numpy` / `numpy_to_tensor` cover the element types the path reads and writes, 4-bit types two per
byte, low nibble first (`core/_pack.py:8-22`; onnx.proto "INT4 / UINT4": the first element in the 4 LSB).
"""
from __future__ import annotations

import os
import struct

import numpy as np

__all__ = ["Message", "parse", "parse_model", "serialize", "load_model", "save_model", "resolve_external_data", "tensor_to_numpy",
           "numpy_to_tensor", "make_attribute", "attribute_value", "make_node", "make_value_info", "DataType", "AttributeType",
           "SCHEMA", "check_model"]


class DataType:
    """TensorProto.DataType codes."""
    UNDEFINED, FLOAT, UINT8, INT8, UINT16, INT16, INT32, INT64, STRING, BOOL, FLOAT16, DOUBLE, UINT32, UINT64 = range(14)
    BFLOAT16 = 16
    UINT4, INT4 = 21, 22


class AttributeType:
    UNDEFINED, FLOAT, INT, STRING, TENSOR, GRAPH, FLOATS, INTS, STRINGS, TENSORS, GRAPHS = range(11)
    TYPE_PROTO, TYPE_PROTOS = 13, 14


_NP_OF = {DataType.FLOAT: np.float32, DataType.UINT8: np.uint8, DataType.INT8: np.int8, DataType.UINT16: np.uint16,
          DataType.INT16: np.int16, DataType.INT32: np.int32, DataType.INT64: np.int64, DataType.BOOL: np.bool_,
          DataType.FLOAT16: np.float16, DataType.DOUBLE: np.float64, DataType.UINT32: np.uint32, DataType.UINT64: np.uint64}
_CODE_OF = {np.dtype(v): k for k, v in _NP_OF.items()}

# message -> {field number: (name, kind, label)}; kind: a scalar kind or the name of a message; label: "one" | "many" |
# "packed".  Transcribed from the published onnx.proto; anything else in a file is carried along as raw bytes.
O, M, P = "one", "many", "packed"
SCHEMA = {
    "ModelProto": {1: ("ir_version", "int64", O), 2: ("producer_name", "string", O), 3: ("producer_version", "string", O),
                   4: ("domain", "string", O), 5: ("model_version", "int64", O), 6: ("doc_string", "string", O),
                   7: ("graph", "GraphProto", O), 8: ("opset_import", "OperatorSetIdProto", M),
                   14: ("metadata_props", "StringStringEntryProto", M), 25: ("functions", "FunctionProto", M)},
    "OperatorSetIdProto": {1: ("domain", "string", O), 2: ("version", "int64", O)},
    "StringStringEntryProto": {1: ("key", "string", O), 2: ("value", "string", O)},
    "GraphProto": {1: ("node", "NodeProto", M), 2: ("name", "string", O), 5: ("initializer", "TensorProto", M),
                   10: ("doc_string", "string", O), 11: ("input", "ValueInfoProto", M), 12: ("output", "ValueInfoProto", M),
                   13: ("value_info", "ValueInfoProto", M), 15: ("sparse_initializer", "SparseTensorProto", M),
                   16: ("metadata_props", "StringStringEntryProto", M)},
    "SparseTensorProto": {1: ("values", "TensorProto", O), 2: ("indices", "TensorProto", O), 3: ("dims", "int64", M)},
    "NodeProto": {1: ("input", "string", M), 2: ("output", "string", M), 3: ("name", "string", O), 4: ("op_type", "string", O),
                  5: ("attribute", "AttributeProto", M), 6: ("doc_string", "string", O), 7: ("domain", "string", O),
                  8: ("overload", "string", O), 9: ("metadata_props", "StringStringEntryProto", M)},
    "AttributeProto": {1: ("name", "string", O), 2: ("f", "float", O), 3: ("i", "int64", O), 4: ("s", "bytes", O),
                       5: ("t", "TensorProto", O), 6: ("g", "GraphProto", O), 7: ("floats", "float", M), 8: ("ints", "int64", M),
                       9: ("strings", "bytes", M), 10: ("tensors", "TensorProto", M), 11: ("graphs", "GraphProto", M),
                       13: ("doc_string", "string", O), 14: ("tp", "TypeProto", O), 15: ("type_protos", "TypeProto", M),
                       20: ("type", "int32", O), 21: ("ref_attr_name", "string", O)},
    "TensorProto": {1: ("dims", "int64", M), 2: ("data_type", "int32", O), 4: ("float_data", "float", P),
                    5: ("int32_data", "int32", P), 6: ("string_data", "bytes", M), 7: ("int64_data", "int64", P),
                    8: ("name", "string", O), 9: ("raw_data", "bytes", O), 10: ("double_data", "double", P),
                    11: ("uint64_data", "uint64", P), 12: ("doc_string", "string", O),
                    13: ("external_data", "StringStringEntryProto", M), 14: ("data_location", "int32", O),
                    16: ("metadata_props", "StringStringEntryProto", M)},
    "ValueInfoProto": {1: ("name", "string", O), 2: ("type", "TypeProto", O), 3: ("doc_string", "string", O),
                       4: ("metadata_props", "StringStringEntryProto", M)},
    "TypeProto": {1: ("tensor_type", "TypeProto.Tensor", O), 6: ("denotation", "string", O)},
    "TypeProto.Tensor": {1: ("elem_type", "int32", O), 2: ("shape", "TensorShapeProto", O)},
    "TensorShapeProto": {1: ("dim", "TensorShapeProto.Dimension", M)},
    "TensorShapeProto.Dimension": {1: ("dim_value", "int64", O), 2: ("dim_param", "string", O), 3: ("denotation", "string", O)},
    "FunctionProto": {1: ("name", "string", O), 4: ("input", "string", M), 5: ("output", "string", M), 6: ("attribute", "string", M),
                      7: ("node", "NodeProto", M), 8: ("doc_string", "string", O), 9: ("opset_import", "OperatorSetIdProto", M),
                      10: ("domain", "string", O), 11: ("attribute_proto", "AttributeProto", M),
                      12: ("value_info", "ValueInfoProto", M), 13: ("overload", "string", O),
                      14: ("metadata_props", "StringStringEntryProto", M)},
}
_BY_NAME = {msg: {name: (num, kind, label) for num, (name, kind, label) in fields.items()} for msg, fields in SCHEMA.items()}
_VARINT_KINDS = {"int64", "int32", "uint64", "bool"}
_WIRE_OF = {"int64": 0, "int32": 0, "uint64": 0, "bool": 0, "double": 1, "float": 5, "string": 2, "bytes": 2}
_FIXED = {"float": ("<f", 4, np.dtype("<f4")), "double": ("<d", 8, np.dtype("<f8"))}


class Message:
    """One protobuf message: the fields of its type as attributes (absent optional field: None; repeated field: a list,
    possibly empty), `_unknown` the records this module has no name for, in file order."""

    __slots__ = ("_type", "_values", "_unknown")

    def __init__(self, type_name: str, **fields):
        if type_name not in SCHEMA:
            raise KeyError(f"onnx_proto: no message type '{type_name}'")
        object.__setattr__(self, "_type", type_name)
        object.__setattr__(self, "_values", {})
        object.__setattr__(self, "_unknown", [])
        for k, v in fields.items():
            setattr(self, k, v)

    def __getattr__(self, name):
        spec = _BY_NAME[self._type].get(name)
        if spec is None:
            raise AttributeError(f"{self._type} has no field '{name}'")
        if name not in self._values:
            if spec[2] == O:
                return None
            self._values[name] = []
        return self._values[name]

    def __setattr__(self, name, value):
        if name not in _BY_NAME[self._type]:
            raise AttributeError(f"{self._type} has no field '{name}'")
        if value is None:
            self._values.pop(name, None)
        else:
            self._values[name] = value

    def has(self, name: str) -> bool:
        v = self._values.get(name)
        return v is not None and not (isinstance(v, list) and not v)

    def copy(self) -> "Message":
        """A copy that shares leaf values (bytes, numbers) but no lists or sub-messages."""
        out = Message(self._type)
        for k, v in self._values.items():
            if isinstance(v, list):
                out._values[k] = [e.copy() if isinstance(e, Message) else e for e in v]
            else:
                out._values[k] = v.copy() if isinstance(v, Message) else v
        out._unknown.extend(self._unknown)
        return out

    def __repr__(self):
        def short(v):
            if isinstance(v, (bytes, memoryview)) and len(v) > 24:
                return f"<{len(v)} bytes>"
            if isinstance(v, list) and len(v) > 6:
                return f"[{len(v)} items]"
            return repr(v)
        return f"{self._type}({', '.join(f'{k}={short(v)}' for k, v in self._values.items())})"


# --------------------------------------------------------------------------------------------------------------- reading
def _varint(buf, pos):
    result = shift = 0
    end = len(buf)
    while True:
        if pos >= end:
            raise ValueError("onnx_proto: truncated varint")
        b = buf[pos]
        pos += 1
        result |= (b & 0x7F) << shift
        if b < 0x80:
            return result, pos
        shift += 7
        if shift > 63:
            raise ValueError("onnx_proto: varint longer than 10 bytes")


def _signed(v: int, kind: str) -> int:
    if kind in ("int64", "int32"):
        v &= 0xFFFFFFFFFFFFFFFF
        return v - (1 << 64) if v >> 63 else v
    return bool(v) if kind == "bool" else v


def _scalar_from_wire(kind, wire, buf, pos):
    if wire != _WIRE_OF.get(kind, wire):                   # the schema says what a field's wire type is: a float sent as varint is malformed input
        raise ValueError(f"onnx_proto: wire type {wire} for a field of kind '{kind}' (the schema says {_WIRE_OF[kind]})")
    if wire == 0:
        v, pos = _varint(buf, pos)
        return _signed(v, kind), pos
    if wire in (1, 5):
        size = 4 if wire == 5 else 8
        if pos + size > len(buf):
            raise ValueError("onnx_proto: truncated fixed-width field")
        return struct.unpack_from("<f" if wire == 5 else "<d", buf, pos)[0], pos + size
    raise ValueError(f"onnx_proto: wire type {wire} for a scalar field")


_MAX_DEPTH = 64          # sub-graphs inside attributes nest; a file that nests deeper than any real model is refused, not recursed into


def parse(type_name: str, data, _depth: int = 0) -> Message:
    """Decode one message of type `type_name` from bytes / memoryview (large byte fields stay views into `data`).  Malformed
    input of any kind is a ValueError."""
    if _depth > _MAX_DEPTH:
        raise ValueError(f"onnx_proto: messages nested deeper than {_MAX_DEPTH}")
    buf = data if isinstance(data, memoryview) else memoryview(data)
    fields = SCHEMA[type_name]
    msg = Message(type_name)
    values = msg._values
    pos, end = 0, len(buf)
    while pos < end:
        key, pos = _varint(buf, pos)
        num, wire = key >> 3, key & 7
        spec = fields.get(num)
        if spec is None:                                   # carried along verbatim
            start = pos
            if wire == 0:
                _, pos = _varint(buf, pos)
            elif wire == 1:
                pos += 8
            elif wire == 5:
                pos += 4
            elif wire == 2:
                n, pos = _varint(buf, pos)
                pos += n
            else:
                raise ValueError(f"onnx_proto: wire type {wire} (groups are not part of onnx.proto)")
            if pos > end:
                raise ValueError("onnx_proto: truncated message")
            msg._unknown.append((key, bytes(buf[start:pos])))
            continue
        name, kind, label = spec
        if wire == 2:
            n, pos = _varint(buf, pos)
            if pos + n > end:
                raise ValueError(f"onnx_proto: field '{name}' of {type_name} runs past the end of its message")
            chunk = buf[pos:pos + n]
            pos += n
            if kind == "string":
                value = bytes(chunk).decode("utf-8", errors="surrogateescape")
            elif kind == "bytes":
                value = chunk if n > 4096 else bytes(chunk)            # raw_data of a large initializer: no copy
            elif kind in SCHEMA:
                value = parse(kind, chunk, _depth + 1)
            else:                                                      # a packed run of scalars
                if label == O:
                    raise ValueError(f"onnx_proto: length-delimited record for scalar field '{name}'")
                if kind in _FIXED:
                    if n % _FIXED[kind][1]:
                        raise ValueError(f"onnx_proto: packed field '{name}' of {type_name} is {n} bytes long")
                    vals = np.frombuffer(chunk, dtype=_FIXED[kind][2]).tolist()
                else:
                    vals, p = [], 0
                    while p < n:
                        v, p = _varint(chunk, p)
                        vals.append(_signed(v, kind))
                values.setdefault(name, []).extend(vals)
                continue
        else:
            if kind in SCHEMA or kind in ("string", "bytes"):
                raise ValueError(f"onnx_proto: wire type {wire} for field '{name}' of {type_name}")
            value, pos = _scalar_from_wire(kind, wire, buf, pos)
            if pos > end:
                raise ValueError("onnx_proto: truncated message")
        if label == O:
            values[name] = value
        else:
            values.setdefault(name, []).append(value)
    return msg


def parse_model(data) -> Message:
    return parse("ModelProto", data)


def load_model(path, load_external_data: bool = True) -> Message:
    """Parse the file at `path`.  Tensors whose bytes live in side files (data_location EXTERNAL: every model past protobuf's
    2 GiB limit) are mapped, not read: `raw_data` becomes a read-only view of an `np.memmap` window, paged in when something
    touches it (`load_external_data=False` leaves the references as they are)."""
    with open(path, "rb") as f:
        model = parse_model(f.read())
    if load_external_data:
        resolve_external_data(model, os.path.dirname(os.path.abspath(path)))
    return model


def _tensors_of(graph):
    yield from graph.initializer
    for n in graph.node:
        for a in n.attribute:
            if a.has("t"):
                yield a.t
            yield from a.tensors
            if a.has("g"):
                yield from _tensors_of(a.g)
            for g in a.graphs:
                yield from _tensors_of(g)


def resolve_external_data(model: Message, base_dir) -> int:
    """Point every externally stored tensor of `model` at a memory map of its file (keys `location`, `offset`, `length` of
    onnx.proto's external_data; the location must stay inside `base_dir`).  Returns the number of tensors mapped."""
    maps: dict = {}
    count = 0
    base_dir = os.path.realpath(base_dir)
    for t in ([] if model.graph is None else _tensors_of(model.graph)):
        if not t.data_location:
            continue
        info = {e.key: e.value for e in t.external_data}
        location = info.get("location")
        if not location:
            raise ValueError(f"onnx_proto: tensor '{t.name}' is external but names no location")
        path = os.path.realpath(os.path.join(base_dir, location))       # symbolic links resolved: the bytes must live inside the directory
        if os.path.commonpath([base_dir, path]) != base_dir:
            raise ValueError(f"onnx_proto: tensor '{t.name}': external data location '{location}' leaves the model's directory")
        if path not in maps:
            maps[path] = np.memmap(path, dtype=np.uint8, mode="r") if os.path.getsize(path) else np.zeros(0, np.uint8)
        offset = int(info.get("offset", 0))
        length = int(info["length"]) if "length" in info else len(maps[path]) - offset
        if offset < 0 or length < 0 or offset + length > len(maps[path]):
            raise ValueError(f"onnx_proto: tensor '{t.name}': bytes [{offset}, {offset + length}) are outside '{location}' "
                             f"({len(maps[path])} bytes)")
        t.raw_data = memoryview(maps[path][offset:offset + length])
        t.external_data = []
        t.data_location = None
        count += 1
    return count


# --------------------------------------------------------------------------------------------------------------- writing
def _put_varint(out: bytearray, v: int) -> None:
    v &= 0xFFFFFFFFFFFFFFFF                                # negative int32 / int64: ten bytes, as protobuf writes them
    while v >= 0x80:
        out.append((v & 0x7F) | 0x80)
        v >>= 7
    out.append(v)


def _put_scalar(out: bytearray, kind: str, v) -> None:
    if kind in _VARINT_KINDS:
        _put_varint(out, int(v))
    else:
        out += struct.pack(_FIXED[kind][0], v)


def _serialize_into(msg: Message, chunks: list) -> int:
    """Appends the encoding of `msg` to `chunks` (bytes-like pieces) and returns its length."""
    fields = SCHEMA[msg._type]
    by_name = _BY_NAME[msg._type]
    total = 0
    for name in sorted(msg._values, key=lambda n: by_name[n][0]):
        value = msg._values[name]
        num, kind, label = by_name[name]
        items = [value] if label == O else value
        if not items:
            continue
        if label == P:
            body = bytearray()
            if kind in _FIXED:
                body += np.asarray(items, dtype=_FIXED[kind][2]).tobytes()
            else:
                for v in items:
                    _put_varint(body, int(v))
            head = bytearray()
            _put_varint(head, (num << 3) | 2)
            _put_varint(head, len(body))
            chunks += [bytes(head), bytes(body)]
            total += len(head) + len(body)
            continue
        for v in items:
            head = bytearray()
            if kind in SCHEMA:
                sub: list = []
                n = _serialize_into(v, sub)
                _put_varint(head, (num << 3) | 2)
                _put_varint(head, n)
                chunks.append(bytes(head))
                chunks += sub
                total += len(head) + n
            elif kind in ("string", "bytes"):
                payload = v.encode("utf-8", errors="surrogateescape") if isinstance(v, str) else v
                _put_varint(head, (num << 3) | 2)
                _put_varint(head, len(payload))
                chunks += [bytes(head), payload]
                total += len(head) + len(payload)
            else:
                _put_varint(head, (num << 3) | _WIRE_OF[kind])
                _put_scalar(head, kind, v)
                chunks.append(bytes(head))
                total += len(head)
    del fields
    for key, raw in msg._unknown:
        head = bytearray()
        _put_varint(head, key)
        chunks += [bytes(head), raw]
        total += len(head) + len(raw)
    return total


def serialize(msg: Message) -> bytes:
    chunks: list = []
    _serialize_into(msg, chunks)
    return b"".join(chunks)


def save_model(model: Message, path, external_data: str | None = None, size_threshold: int = 1024) -> None:
    """Write `model` to `path`.  With `external_data` (a file name next to `path`, e.g. "model.onnx.data") every tensor of at
    least `size_threshold` bytes goes to that file -- offsets aligned to 4096 bytes for tensors of a megabyte and more (what
    memory-mapping loaders ask for), 64 otherwise -- and the model file keeps the references.  Without it everything is
    inline, which protobuf caps at 2 GiB: a larger model is refused rather than written unreadable.  `model` itself is not
    modified."""
    path = os.fspath(path)
    model_tmp = f"{path}.tmp{os.getpid()}"                 # both files are written under temporary names and renamed when complete:
    if external_data is None:                              # a reader never sees half a file, and a model whose tensors are mapped
        chunks: list = []                                  # from the very files being replaced (dst == src) keeps its old bytes
        n = _serialize_into(model, chunks)
        if n >= 1 << 31:
            raise ValueError(f"onnx_proto: the serialised model is {n} bytes; a protobuf message cannot exceed 2 GiB: pass "
                             "external_data='<file name>' to keep the tensors in a side file")
        try:
            with open(model_tmp, "wb") as f:
                for c in chunks:
                    f.write(c)
            os.replace(model_tmp, path)
        finally:
            _remove_quietly(model_tmp)
        return
    if os.path.basename(external_data) != external_data:
        raise ValueError("onnx_proto: external_data is a file name next to the model, not a path")
    base = os.path.dirname(os.path.abspath(path))
    data_path = os.path.join(base, external_data)
    data_tmp = f"{data_path}.tmp{os.getpid()}"
    swapped = []                                           # (tensor, raw_data) put back when the file is written
    try:
        with open(data_tmp, "wb") as data:
            pos = 0
            for t in _tensors_of(model.graph):
                if t.data_location or not t.has("raw_data") or len(t.raw_data) < size_threshold:
                    continue
                raw = t.raw_data
                align = 4096 if len(raw) >= 1 << 20 else 64
                pad = -pos % align
                if pad:
                    data.write(b"\0" * pad)
                    pos += pad
                data.write(raw)
                swapped.append((t, raw))
                t.raw_data = None
                t.data_location = 1
                t.external_data = [Message("StringStringEntryProto", key=k, value=v)
                                   for k, v in (("location", external_data), ("offset", str(pos)), ("length", str(len(raw))))]
                pos += len(raw)
        chunks = []
        n = _serialize_into(model, chunks)
        if n >= 1 << 31:
            raise ValueError(f"onnx_proto: {n} bytes remain inline; lower size_threshold")
        with open(model_tmp, "wb") as f:
            for c in chunks:
                f.write(c)
        os.replace(data_tmp, data_path)
        os.replace(model_tmp, path)
    finally:
        for t, raw in swapped:
            t.raw_data, t.data_location, t.external_data = raw, None, []
        _remove_quietly(data_tmp)
        _remove_quietly(model_tmp)


def _remove_quietly(path) -> None:
    try:
        os.unlink(path)
    except OSError:
        pass


# --------------------------------------------------------------------------------------------------------------- tensors
def _unpack_nibbles(raw: np.ndarray, count: int, signed: bool) -> np.ndarray:
    lo, hi = raw & 0x0F, raw >> 4
    out = np.empty(raw.size * 2, dtype=np.uint8)
    out[0::2], out[1::2] = lo, hi
    out = out[:count]
    if signed:
        return ((out ^ 8).astype(np.int16) - 8).astype(np.int8)      # two's-complement nibble -> int8
    return out


def tensor_to_numpy(t: Message) -> np.ndarray:
    """The array a TensorProto holds (raw_data or the typed fields).  4-bit types come back one value per byte (int8 /
    uint8 containers).  Tensors whose bytes live in another file (data_location EXTERNAL) are refused: the caller decides
    where those are."""
    if t.data_location:
        raise ValueError(f"onnx_proto: tensor '{t.name}' keeps its data in an external file")
    dims = [int(d) for d in t.dims]
    count = int(np.prod(dims, dtype=np.int64)) if dims else 1
    code = t.data_type
    if code in (DataType.UINT4, DataType.INT4):
        if t.has("raw_data"):
            raw = np.frombuffer(t.raw_data, dtype=np.uint8)
        else:                                                         # int32_data: one packed byte per entry
            raw = np.asarray(t.int32_data, dtype=np.int64).astype(np.uint8)
        if raw.size != (count + 1) // 2:
            raise ValueError(f"onnx_proto: tensor '{t.name}': {raw.size} bytes for {count} 4-bit values")
        return _unpack_nibbles(raw, count, code == DataType.INT4).reshape(dims)
    if code == DataType.BFLOAT16:
        raise ValueError(f"onnx_proto: tensor '{t.name}' is bfloat16 (no NumPy type here)")
    if code not in _NP_OF:
        raise ValueError(f"onnx_proto: tensor '{t.name}' has element type {code}, which this module does not read")
    dt = np.dtype(_NP_OF[code])
    if t.has("raw_data"):
        a = np.frombuffer(t.raw_data, dtype=dt.newbyteorder("<"))
    elif code == DataType.FLOAT:
        a = np.asarray(t.float_data, dtype=np.float32)
    elif code == DataType.DOUBLE:
        a = np.asarray(t.double_data, dtype=np.float64)
    elif code == DataType.INT64:
        a = np.asarray(t.int64_data, dtype=np.int64)
    elif code in (DataType.UINT32, DataType.UINT64):
        a = np.asarray(t.uint64_data, dtype=np.uint64).astype(dt)
    elif code == DataType.FLOAT16:                                     # bit patterns in int32_data
        a = np.asarray(t.int32_data, dtype=np.int64).astype(np.uint16).view(np.float16)
    else:                                                              # the other narrow types travel in int32_data
        a = np.asarray(t.int32_data, dtype=np.int64).astype(dt)
    if a.size != count:
        raise ValueError(f"onnx_proto: tensor '{t.name}': {a.size} values for shape {dims}")
    return a.reshape(dims)


def numpy_to_tensor(name: str, array, data_type: int | None = None) -> Message:
    """A TensorProto with raw_data (what `ir.tensor(...)` serialises to).  `data_type` UINT4 / INT4 packs an int8 / uint8
    container two values per byte, low nibble first, padding an odd count with a zero nibble."""
    a = np.asarray(array)
    if data_type is None:
        if a.dtype not in _CODE_OF:
            raise ValueError(f"onnx_proto: no ONNX element type for NumPy dtype {a.dtype}")
        data_type = _CODE_OF[a.dtype]
    t = Message("TensorProto", dims=[int(d) for d in a.shape], data_type=int(data_type), name=name)
    if data_type in (DataType.UINT4, DataType.INT4):
        if a.dtype not in (np.dtype(np.int8), np.dtype(np.uint8)):
            raise ValueError(f"onnx_proto: tensor '{name}': 4-bit values travel in int8 / uint8 containers, not {a.dtype}")
        u = np.ascontiguousarray(a).reshape(-1).view(np.uint8)          # two's-complement nibbles: the low four bits as they are
        lo_ok, hi_ok = (-8, 7) if data_type == DataType.INT4 else (0, 15)
        bad = ((u + np.uint8(8)) & np.uint8(0xF0)).any() if data_type == DataType.INT4 and a.dtype == np.int8 else (u & np.uint8(0xF0)).any()
        if bad:
            raise ValueError(f"onnx_proto: tensor '{name}' holds values outside the 4-bit range [{lo_ok}, {hi_ok}]")
        if u.size % 2:
            u = np.concatenate([u, np.zeros(1, dtype=np.uint8)])
        t.raw_data = ((u[0::2] & np.uint8(0x0F)) | (u[1::2] << np.uint8(4))).tobytes()
        return t
    want = np.dtype(_NP_OF[data_type])
    if a.dtype != want:
        raise ValueError(f"onnx_proto: tensor '{name}': array dtype {a.dtype} does not match element type {data_type}")
    body = np.ascontiguousarray(a).astype(want.newbyteorder("<"), copy=False).reshape(-1)
    # a large array is referenced, not copied (the memoryview keeps it alive; writers stream it out as it is)
    t.raw_data = memoryview(body).cast("B") if body.nbytes >= 1 << 20 else body.tobytes()
    return t


# --------------------------------------------------------------------------------------------------------------- checking
_ITEM_BITS = {DataType.FLOAT: 32, DataType.UINT8: 8, DataType.INT8: 8, DataType.UINT16: 16, DataType.INT16: 16, DataType.INT32: 32,
              DataType.INT64: 64, DataType.BOOL: 8, DataType.FLOAT16: 16, DataType.DOUBLE: 64, DataType.UINT32: 32, DataType.UINT64: 64,
              DataType.BFLOAT16: 16, DataType.UINT4: 4, DataType.INT4: 4}


def _check_tensor(t: Message, where: str) -> None:
    if not t.name:
        raise ValueError(f"check_model: {where}: a tensor without a name")
    if any(int(d) < 0 for d in t.dims):
        raise ValueError(f"check_model: tensor '{t.name}' has a negative dimension")
    bits = _ITEM_BITS.get(t.data_type)
    if bits is not None and t.has("raw_data") and not t.data_location:
        count = 1
        for d in t.dims:
            count *= int(d)
        want = (count * bits + 7) // 8
        if len(t.raw_data) != want:
            raise ValueError(f"check_model: tensor '{t.name}': {len(t.raw_data)} bytes of raw data for shape {list(t.dims)} "
                             f"of element type {t.data_type} ({want} expected)")


def _check_graph(graph: Message, outer: set, imported: set, functions: set, where: str) -> None:
    known = set(outer)
    names = set()
    for t in graph.initializer:
        _check_tensor(t, where)
        if t.name in names:
            raise ValueError(f"check_model: {where}: two initializers named '{t.name}'")
        names.add(t.name)
    known |= names | {i.name for i in graph.input} | {t.values.name for t in graph.sparse_initializer if t.values is not None}
    produced = set()
    for n in graph.node:
        label = f"{where}: node '{n.name or n.op_type}'"
        if not n.op_type:
            raise ValueError(f"check_model: {label} has no operator type")
        domain = "" if n.domain in (None, "ai.onnx") else n.domain
        if domain not in imported:
            raise ValueError(f"check_model: {label} uses domain '{domain}', which the model does not import")
        if domain == "quant" and (domain, n.op_type, n.overload or "") not in functions:
            raise ValueError(f"check_model: {label} calls {domain}::{n.op_type}, which the model does not define")
        for v in n.input:
            if v and v not in known:
                raise ValueError(f"check_model: {label} reads '{v}' before anything produces it")
        for a in n.attribute:
            for g in ([a.g] if a.has("g") else []) + list(a.graphs):
                _check_graph(g, known, imported, functions, f"{label} / sub-graph")
            for t in ([a.t] if a.has("t") else []):
                if t.name:
                    _check_tensor(t, label)
        for o in n.output:
            if o:
                if o in produced or o in names:
                    raise ValueError(f"check_model: {label} writes '{o}', which already has a producer")
                produced.add(o)
                known.add(o)
    for o in graph.output:
        if o.name not in known:
            raise ValueError(f"check_model: {where}: graph output '{o.name}' is never produced")


def check_model(model: Message) -> None:
    """The structural part of `onnx.checker.check_model` (the package is not here): an IR version and a default opset, unique
    initializers whose byte counts fit their shapes, nodes in topological order in SSA form, every operator domain imported,
    every `quant`-domain call defined by a function of the model, closed and well-formed function bodies, graph outputs
    produced.  Raises ValueError naming the first offender."""
    if model._type != "ModelProto" or model.graph is None:
        raise ValueError("check_model: not a model with a graph")
    if not model.ir_version:
        raise ValueError("check_model: no ir_version")
    imported = {("" if o.domain in (None, "ai.onnx") else o.domain) for o in model.opset_import}
    if "" not in imported:
        raise ValueError("check_model: the default operator set is not imported")
    functions = {(f.domain or "", f.name, f.overload or "") for f in model.functions}
    if len(functions) != len(model.functions):
        raise ValueError("check_model: two functions with the same domain, name and overload")
    for f in model.functions:
        f_imports = {("" if o.domain in (None, "ai.onnx") else o.domain) for o in f.opset_import}
        body = Message("GraphProto", node=list(f.node), input=[Message("ValueInfoProto", name=i) for i in f.input],
                       output=[Message("ValueInfoProto", name=o) for o in f.output])
        _check_graph(body, set(), f_imports, functions, f"function {f.domain}::{f.name}")
    _check_graph(model.graph, set(), imported, functions, "graph")


# --------------------------------------------------------------------------------------------------------------- helpers
def make_attribute(name: str, value) -> Message:
    """int / float / str / bytes / Message(TensorProto | GraphProto) and homogeneous lists of int / float / str."""
    a = Message("AttributeProto", name=name)
    if isinstance(value, bool) or isinstance(value, (int, np.integer)):
        a.i, a.type = int(value), AttributeType.INT
    elif isinstance(value, (float, np.floating)):
        a.f, a.type = float(value), AttributeType.FLOAT
    elif isinstance(value, str):
        a.s, a.type = value.encode("utf-8"), AttributeType.STRING
    elif isinstance(value, bytes):
        a.s, a.type = value, AttributeType.STRING
    elif isinstance(value, Message) and value._type == "TensorProto":
        a.t, a.type = value, AttributeType.TENSOR
    elif isinstance(value, Message) and value._type == "GraphProto":
        a.g, a.type = value, AttributeType.GRAPH
    elif isinstance(value, (list, tuple)):
        if all(isinstance(v, (int, np.integer)) and not isinstance(v, bool) for v in value):
            a.ints, a.type = [int(v) for v in value], AttributeType.INTS
        elif all(isinstance(v, (int, float, np.integer, np.floating)) for v in value):
            a.floats, a.type = [float(v) for v in value], AttributeType.FLOATS
        elif all(isinstance(v, (str, bytes)) for v in value):
            a.strings, a.type = [v.encode("utf-8") if isinstance(v, str) else v for v in value], AttributeType.STRINGS
        else:
            raise TypeError(f"onnx_proto: attribute '{name}': a list of mixed types")
    else:
        raise TypeError(f"onnx_proto: attribute '{name}': unsupported value {type(value)}")
    return a


def attribute_value(a: Message):
    """The Python value of an AttributeProto (strings as str; tensors as NumPy arrays; graphs as messages)."""
    t = a.type
    if t == AttributeType.INT:
        return int(a.i or 0)
    if t == AttributeType.FLOAT:
        return float(a.f or 0.0)
    if t == AttributeType.STRING:
        return bytes(a.s or b"").decode("utf-8", errors="surrogateescape")
    if t == AttributeType.TENSOR:
        return tensor_to_numpy(a.t)
    if t == AttributeType.GRAPH:
        return a.g
    if t == AttributeType.INTS:
        return [int(v) for v in a.ints]
    if t == AttributeType.FLOATS:
        return [float(v) for v in a.floats]
    if t == AttributeType.STRINGS:
        return [bytes(s).decode("utf-8", errors="surrogateescape") for s in a.strings]
    if t in (None, AttributeType.UNDEFINED):                 # writers before IR 3 left `type` out: go by what is set
        for field in ("i", "f", "s", "t", "g"):
            if a.has(field):
                v = getattr(a, field)
                return bytes(v).decode("utf-8", errors="surrogateescape") if field == "s" else (tensor_to_numpy(v) if field == "t" else v)
        for field in ("ints", "floats", "strings"):
            if a.has(field):
                return list(getattr(a, field))
    raise ValueError(f"onnx_proto: attribute '{a.name}' of type {t} is not read by this module")


def make_node(op_type: str, inputs, outputs, name: str | None = None, domain: str | None = None, **attrs) -> Message:
    n = Message("NodeProto", input=[("" if i is None else i) for i in inputs], output=list(outputs), op_type=op_type)
    if name:
        n.name = name
    if domain:
        n.domain = domain
    if attrs:
        n.attribute = [make_attribute(k, v) for k, v in attrs.items()]
    return n


def make_value_info(name: str, elem_type: int, shape) -> Message:
    """`shape`: ints, strs (symbolic) or None entries; None as the shape itself leaves the rank open."""
    tt = Message("TypeProto.Tensor", elem_type=int(elem_type))
    if shape is not None:
        dims = []
        for d in shape:
            dim = Message("TensorShapeProto.Dimension")
            if isinstance(d, (int, np.integer)):
                dim.dim_value = int(d)
            elif isinstance(d, str):
                dim.dim_param = d
            dims.append(dim)
        tt.shape = Message("TensorShapeProto", dim=dims)
    return Message("ValueInfoProto", name=name, type=Message("TypeProto", tensor_type=tt))
