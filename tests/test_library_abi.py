"""The C-ABI library: builds for gfx950 without a GPU, loads, and exports exactly the symbols that
include/oq_hip.h declares and the ctypes stub binds.  No compute call is made here (CPU-only suite)."""
import ctypes
import os
import re
import subprocess

import pytest

from conftest import ROOT

HEADER = os.path.join(ROOT, "include", "oq_hip.h")


def declared_symbols():
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(oq_[a-z0-9_]+)\s*\(", text)))


@pytest.fixture(scope="module")
def lib_path():
    from onnx_quantize_amd import _build
    return _build.build(verbose=False)


def test_header_matches_ctypes_prototypes():
    from onnx_quantize_amd.hip import _lib
    assert declared_symbols() == sorted(_lib.PROTOTYPES)


def test_library_exports_every_declared_symbol(lib_path):
    lib = ctypes.CDLL(lib_path)
    for name in declared_symbols():
        assert hasattr(lib, name), f"{name} is declared in include/oq_hip.h but not exported"
    lib.oq_target_arch.restype = ctypes.c_char_p
    assert lib.oq_target_arch() == b"gfx950"
    assert lib.oq_abi_version() == 1


def test_code_object_is_gfx950(lib_path, tmp_path):
    # llvm-objdump --offloading drops the extracted code objects next to its input: work on a copy outside the tree
    import shutil
    copy = shutil.copy(lib_path, tmp_path / "liboq_hip.so")
    out = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-objdump", "--offloading", str(copy)], capture_output=True, text=True,
                         cwd=tmp_path)
    assert "gfx950" in out.stdout + out.stderr


def test_header_is_plain_c_and_links(lib_path, tmp_path):
    """tests/c/abi_check.c: include/oq_hip.h compiles as C99, every declared entry point resolves against the library
    and the host-only calls answer; the list of symbols in the C file must be the header's."""
    src = os.path.join(ROOT, "tests", "c", "abi_check.c")
    text = open(src).read()
    assert sorted(set(re.findall(r"TAKE\((oq_[a-z0-9_]+)\)", text))) == declared_symbols()
    exe = tmp_path / "abi_check"
    libdir = os.path.dirname(lib_path)
    cc = subprocess.run(["gcc", "-std=c99", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), src, "-o", str(exe),
                         "-L", libdir, "-loq_hip", f"-Wl,-rpath,{libdir}", "-Wl,-rpath,/opt/rocm/lib"],
                        capture_output=True, text=True)
    assert cc.returncode == 0, cc.stdout + cc.stderr
    env = dict(os.environ)
    import torch
    env["LD_LIBRARY_PATH"] = os.pathsep.join([os.path.join(os.path.dirname(torch.__file__), "lib"), "/opt/rocm/lib",
                                               env.get("LD_LIBRARY_PATH", "")])
    run = subprocess.run([str(exe)], capture_output=True, text=True, env=env)
    assert run.returncode == 0 and run.stdout.startswith("ok "), (run.returncode, run.stdout, run.stderr)


def test_host_only_entry_points(lib_path):
    """oq_qrange / status strings / argument validation run on the host and need no device."""
    from onnx_quantize_amd.hip import _lib
    lib = _lib.load()
    assert _lib.qrange(_lib.OQ_INT4, True, False) == (-7, 7)
    assert _lib.qrange(_lib.OQ_UINT8, True, True) == (0, 127)
    assert _lib.qrange(_lib.OQ_INT32, False, False) == (-2**31, 2**31 - 1)
    assert lib.oq_status_string(-3) == b"workspace too small"
    with pytest.raises(_lib.OqHipError, match="unknown quantization type"):
        _lib.qrange(17, False, False)
    # null pointers are rejected before any launch
    st = lib.oq_rtn_quantize_f32(None, 4, 4, 4, 0, 0, -1, 0, 0, 1.0, 0, None, None, None, 0, None, 0, None)
    assert st == -1 and b"null pointer" in lib.oq_last_error()
    assert lib.oq_rtn_workspace_bytes(4096, 11008, _lib.OQ_GROUP, 128, 0) > 0


def test_missing_library_is_loud(monkeypatch, tmp_path):
    from onnx_quantize_amd.hip import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(_lib.OqHipMissing, match="no CPU fallback"):
        _lib.load()


def test_ops_refuse_host_tensors():
    import torch
    from onnx_quantize_amd.hip import ops
    with pytest.raises(TypeError, match="no CPU fallback"):
        ops.rtn_quantize(torch.zeros(4, 4), "int8", "tensor")
    with pytest.raises(TypeError):
        ops.minmax_collect(torch.zeros(4), torch.zeros(4), 0.0)
