"""Occupancy guard for the two headline RTN kernels.  Both sit right at a register boundary of the gfx950 allocation table
(MI355X_MICROARCH.md, Register files): `rtn_group_wave<8, true, 5>` needs <= 96 VGPRs for five waves per SIMD (92 today; the
plain build has 100), `rtn_group_fused<16, ...>` needs <= 128 for four (126 today).  Four registers more in the fused
kernel cost 30 % of its time in round 2 (44.6 -> 58.5 us) without a single test failing, so the compiler's own resource
report is checked here (hipcc cross-compiles without a GPU)."""
import os
import re
import shutil
import subprocess

import pytest

from conftest import ROOT

HIPCC = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not available")
def test_headline_kernels_keep_their_occupancy(tmp_path):
    from onnx_quantize_amd import _build
    src = os.path.join(ROOT, "onnx_quantize_amd", "csrc", "rtn.hip")
    r = subprocess.run([HIPCC, *_build.CXXFLAGS, "-S", "--cuda-device-only", "-Rpass-analysis=kernel-resource-usage", "-o",
                        str(tmp_path / "rtn.s"), src], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    report = {}
    for m in re.finditer(r"Function Name: (\S+).*?VGPRs: (\d+).*?Occupancy \[waves/SIMD\]: (\d+).*?VGPRs Spill: (\d+)", r.stderr, re.S):
        report[m.group(1)] = tuple(int(m.group(i)) for i in (2, 3, 4))
    wave = report["_ZN2oq14rtn_group_waveILi8ELb1ELi5EEEvNS_7RtnArgsE"]            # MatMulNBits blob, g = 128: the headline launch
    fused = report["_ZN2oq15rtn_group_fusedILi16ELb1ELb1ELb1EEEvNS_7RtnArgsE"]     # [K, N] bytes, g = 128
    assert wave[0] <= 96 and wave[1] >= 5 and wave[2] == 0, wave
    assert fused[0] <= 128 and fused[1] >= 4 and fused[2] == 0, fused
    for name, (vgprs, occ, spill) in report.items():
        assert spill == 0, (name, vgprs, occ, spill)                                   # no kernel of the file may spill
