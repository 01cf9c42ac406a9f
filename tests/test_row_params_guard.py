"""CPU guard for a GPU abort fixed in round 2 (VERDICT r02 / ADVICE r02): `elementwise.hip::tile_kernel` loaded the parameter
row one entry past the end of scale / zp after a tile's last row.  The walk now lives in csrc/row_params.hpp, shared by
the kernel and by tests/c/row_params_walk.cpp, which this test builds with AddressSanitizer and runs on the host."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "onnx_quantize_amd", "csrc")


def _build(tmp_path, extra=()):
    gxx = shutil.which("g++")
    if gxx is None:
        pytest.skip("g++ not available")
    exe = str(tmp_path / "row_params_walk")
    subprocess.run([gxx, "-std=c++17", "-O1", "-g", "-Wall", "-Werror", "-fsanitize=address,undefined", "-fno-sanitize-recover=all",
                    "-I", CSRC, *extra, os.path.join(ROOT, "tests", "c", "row_params_walk.cpp"), "-o", exe],
                   check=True, capture_output=True, text=True)
    return exe


def test_tile_parameter_walk_stays_inside_the_arrays_under_asan(tmp_path):
    exe = _build(tmp_path)
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    assert "bad=0" in r.stdout


def test_the_guard_catches_the_round_2_bug(tmp_path):
    """The same program with the cursor's `another row follows` condition removed (the round-2 bug) must die under ASan --
    otherwise the guard guards nothing."""
    src = open(os.path.join(CSRC, "row_params.hpp")).read()
    broken = src.replace("if (rp.next() && r + 1 < r_end) load(rp.base);", "if (rp.next()) load(rp.base);")
    assert broken != src
    inc = tmp_path / "inc"
    inc.mkdir()
    (inc / "row_params.hpp").write_text(broken)
    gxx = shutil.which("g++")
    if gxx is None:
        pytest.skip("g++ not available")
    exe = str(tmp_path / "broken")
    subprocess.run([gxx, "-std=c++17", "-O1", "-g", "-fsanitize=address", "-I", str(inc), os.path.join(ROOT, "tests", "c", "row_params_walk.cpp"),
                    "-o", exe], check=True, capture_output=True, text=True)
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "heap-buffer-overflow" in r.stderr


def test_the_kernel_uses_the_shared_walk():
    src = open(os.path.join(CSRC, "elementwise.hip")).read()
    assert '#include "row_params.hpp"' in src and "TileParamCursor<" in src and "params.row_done(" in src
    assert "struct RowParams" not in src      # one definition only: the header's
