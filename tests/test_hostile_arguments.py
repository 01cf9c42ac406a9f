"""Hostile arguments through the C ABI (CPU suite: argument validation runs on the host, before any launch).

Every entry point of include/oq_hip.h that takes pointers is called with null pointers, zero / negative / absurdly large
extents (alone and all together), unknown enum codes and empty workspaces.  The contract (include/oq_hip.h, "Errors"): a
negative status and a message, never a signal -- an integer overflow in `K * N`, a division by an overflowed product or a grid
dimension truncated to 32 bits would be a fault or a silently shortened launch on a GPU box.  Without a device a call that
passes validation ends in OQ_ERR_LAUNCH (-4, "no ROCm-capable device"): that is how this file tells "refused" from "would have
been launched", and an extent beyond the documented bounds must never get that far.

Each entry point runs in its own process, so that a crash names its function.
"""
import ctypes as C
import json
import os
import re
import subprocess
import sys

import pytest

from conftest import ROOT

HEADER = os.path.join(ROOT, "include", "oq_hip.h")
# parameters that are extents of an operand (bounded by the ABI) -- as opposed to byte counts, strides of caller-owned batches,
# running sample counts and the like, where a large value is legitimate
EXTENTS = {"K", "N", "T", "R", "C", "ldw", "ldx", "ldo", "count", "n", "batch", "blocks", "block_size"}


def prototypes_with_names():
    """name -> [(ctype text, parameter name)] parsed from the header (comments stripped)."""
    text = re.sub(r"/\*.*?\*/", "", open(HEADER).read(), flags=re.S)
    text = re.sub(r"//[^\n]*", "", text)
    out = {}
    for m in re.finditer(r"\b(?:int32_t|size_t)\s+(oq_[a-z0-9_]+)\s*\(([^;{]*?)\)\s*;", text, flags=re.S):
        params = []
        for p in m.group(2).split(","):
            p = " ".join(p.split())
            if not p or p == "void":
                continue
            pm = re.match(r"(.*?)(\w+)$", p)
            params.append((pm.group(1).strip(), pm.group(2)))
        out[m.group(1)] = params
    return out


def functions_under_test():
    from onnx_quantize_amd.hip import _lib
    return sorted(n for n, (res, args) in _lib.PROTOTYPES.items()
                  if res is C.c_int32 and any(t is C.c_void_p or t is C.POINTER(C.c_int64) for t in args))


CHILD = r"""
import ctypes as C, json, sys
sys.path.insert(0, sys.argv[1])
from onnx_quantize_amd.hip import _lib
name = sys.argv[2]
names = json.loads(sys.argv[3])
lib = _lib.load()
res, argtypes = _lib.PROTOTYPES[name]
buf = (C.c_char * 65536)()            # zero-filled host memory standing in for every pointer (never a device pointer: nothing
ptr = C.cast(buf, C.c_void_p)         # may be launched on it, and host-read tables in it hold zeros = invalid entries)

def build(base_int, ptr_value, size_value, overrides=()):
    args = []
    for i, t in enumerate(argtypes):
        if t is C.c_void_p:
            args.append(ptr_value)
        elif t is C.POINTER(C.c_int64):
            args.append(C.cast(buf, C.POINTER(C.c_int64)) if ptr_value is not None else None)
        elif t in (C.c_float, C.c_double):
            args.append(0.5)
        elif t is C.c_size_t:
            args.append(size_value)
        else:
            args.append(base_int)
    for i, v in overrides:
        args[i] = v
    return args

out = []
def call(tag, args):
    st = getattr(lib, name)(*args)
    out.append([tag, int(st), lib.oq_last_error().decode(errors="replace")[:160]])

call("null", build(4, None, 0))
for v in (0, -1, 1 << 41, (1 << 62) + 12345):
    call("all=%d" % v, build(v, ptr, 0))
    call("all=%d+ws" % v, build(v, ptr, 1 << 62))
call("enums=99", build(16, ptr, 1 << 20, [(i, 99) for i, t in enumerate(argtypes) if t is C.c_int32]))
for i, t in enumerate(argtypes):
    if t is C.c_int64:
        for v in (1 << 41, (1 << 62) + 12345, -(1 << 62)):
            call("%s=%d" % (names[i], v), build(16, ptr, 1 << 62, [(i, v)]))
print(json.dumps(out))
"""


@pytest.mark.parametrize("name", functions_under_test())
def test_entry_point_refuses_hostile_arguments(name, tmp_path):
    from onnx_quantize_amd import _build
    _build.build(verbose=False)
    protos = prototypes_with_names()
    assert name in protos, name
    names = [p[1] for p in protos[name]]
    script = tmp_path / "child.py"
    script.write_text(CHILD)
    run = subprocess.run([sys.executable, str(script), ROOT, name, json.dumps(names)], capture_output=True, text=True, timeout=300)
    assert run.returncode == 0, f"{name} died with {run.returncode} (negative = signal): {run.stderr[-400:]}"
    results = json.loads(run.stdout.strip().splitlines()[-1])
    for tag, status, message in results:
        if name == "oq_qrange" and status == 0:
            continue                                   # a valid type code and two valid output pointers: a legitimate success
        assert status < 0, (name, tag, status, message)
        assert message, (name, tag)                    # a refused call always says why
        param = tag.split("=")[0]
        huge = tag.startswith("all=2199023255552") or tag.startswith("all=4611686018427400249") or \
            (param in EXTENTS and "=" in tag and not tag.startswith("all=") and abs(int(tag.split("=")[1].split("+")[0])) >= 1 << 41)
        if huge:
            # beyond the bounds of the ABI: refused by validation (-1 / -2 / -3), never handed to a launch (-4 here, a fault there)
            assert status in (-1, -2, -3), (name, tag, status, message)
