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
    r = subprocess.run([HIPCC, *_build.flags_for(src), "-S", "--cuda-device-only", "-Rpass-analysis=kernel-resource-usage", "-o",
                        str(tmp_path / "rtn.s"), src], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    report = {}
    for m in re.finditer(r"Function Name: (\S+).*?VGPRs: (\d+).*?Occupancy \[waves/SIMD\]: (\d+).*?VGPRs Spill: (\d+)", r.stderr, re.S):
        report[m.group(1)] = tuple(int(m.group(i)) for i in (2, 3, 4))
    wave = report["_ZN2oq14rtn_group_waveILi8ELb1ELi5EEEvNS_7RtnArgsE"]            # MatMulNBits blob, g = 128: the headline launch
    fused = report["_ZN2oq15rtn_group_fusedILi16ELb1ELb1ELb1ELb0EEEvNS_7RtnArgsE"]     # [K, N] bytes, g = 128
    fused_tr = report["_ZN2oq15rtn_group_fusedILi16ELb1ELb1ELb1ELb1EEEvNS_7RtnArgsE"]  # the same with the parameters transposed inside the launch (round 6)
    assert fused_tr[0] <= 128 and fused_tr[1] >= 4 and fused_tr[2] == 0, fused_tr
    assert wave[0] <= 96 and wave[1] >= 5 and wave[2] == 0, wave
    assert fused[0] <= 128 and fused[1] >= 4 and fused[2] == 0, fused
    for name, (vgprs, occ, spill) in report.items():
        assert spill == 0, (name, vgprs, occ, spill)                                   # no kernel of the file may spill


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not available")
def test_hessian_gemm_kernels_fit_two_waves_per_simd_without_spilling(tmp_path):
    """The split-operand SYRK kernels run one 8-wave block per CU (two waves per SIMD: <= 256 registers) with all 128
    accumulator registers live across the stage loop: a spill there would sit inside the MFMA stream."""
    from onnx_quantize_amd import _build
    src = os.path.join(ROOT, "onnx_quantize_amd", "csrc", "syrk_bf16x3.hip")
    r = subprocess.run([HIPCC, *_build.flags_for(src), "-S", "--cuda-device-only", "-Rpass-analysis=kernel-resource-usage", "-o",
                        str(tmp_path / "syrk.s"), src], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    seen = 0
    for m in re.finditer(r"Function Name: (\S+).*?VGPRs: (\d+).*?ScratchSize \[bytes/lane\]: (\d+).*?Occupancy \[waves/SIMD\]: (\d+)", r.stderr, re.S):
        name, vgprs, scratch, occ = m.group(1), int(m.group(2)), int(m.group(3)), int(m.group(4))
        if any(k in name for k in ("syrk_pieces_kernel", "syrk_f16_m16_kernel", "syrk_f16_m16_many_kernel", "gemm_f16x3_kernel", "gemm_f16x3_many_kernel")):
            seen += 1
            assert vgprs <= 256 and scratch == 0 and occ >= 2, (name, vgprs, scratch, occ)
    # three instantiations of the 32x32 form + the 16x16x32 fp16 kernel and its many-item form + the two-operand GEMM's two
    # epilogues, the one-product loss form and the dot-product forms with three and two products (AWQ searches) and its many-problem form
    assert seen == 11


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not available")
def test_gptq_loop_kernels_do_not_spill(tmp_path):
    """`gptq_rows16_kernel` is launched with up to 1024 threads (<= 128 registers) and is a chain of dependent instructions:
    twice during its writing the optimiser produced 140 spilled registers (updates of later slabs sunk to the end of the
    slab; all 16 steps' LDS reads hoisted) without any functional test noticing.  `panel_update_kernel` needs two blocks per
    CU (64 KB of LDS each)."""
    from onnx_quantize_amd import _build
    src = os.path.join(ROOT, "onnx_quantize_amd", "csrc", "gptq_loop.hip")
    r = subprocess.run([HIPCC, *_build.flags_for(src), "-S", "--cuda-device-only", "-Rpass-analysis=kernel-resource-usage", "-o",
                        str(tmp_path / "loop.s"), src], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    seen = {}
    for m in re.finditer(r"Function Name: (\S+).*?VGPRs: (\d+).*?ScratchSize \[bytes/lane\]: (\d+).*?Occupancy \[waves/SIMD\]: (\d+).*?LDS Size \[bytes/block\]: (\d+)",
                         r.stderr, re.S):
        seen[m.group(1)] = tuple(int(m.group(i)) for i in (2, 3, 4, 5))
    rows16 = next(v for k, v in seen.items() if "gptq_rows16_kernel" in k)
    panel = next(v for k, v in seen.items() if "panel_update_kernel" in k)
    assert rows16[0] <= 128 and rows16[1] == 0 and rows16[3] <= 65536, rows16
    assert panel[1] == 0 and panel[2] >= 2 and panel[3] <= 65536, panel


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not available")
def test_resident_rtn_kernels_keep_their_tiles_in_registers(tmp_path):
    """`rtn_resident_groups` (channel, tall groups: W read once) holds a 128 x 256 tile in 64 registers per lane while the range
    completes elsewhere; two 8-wave workgroups per CU (<= 128 registers, no scratch) are what keeps loads in flight while one of
    them waits.  `rtn_tensor_onepass` runs ONE 8-wave workgroup per CU that keeps a tile in 64 architectural registers, two in
    its 128 accumulation registers -- named as PHYSICAL registers a[0..127] in the assembly text, so the register allocator
    must not have placed anything of its own there -- and one in LDS; a spill would turn its kept tiles into scratch traffic."""
    from onnx_quantize_amd import _build
    src = os.path.join(ROOT, "onnx_quantize_amd", "csrc", "rtn_resident.hip")
    r = subprocess.run([HIPCC, *_build.flags_for(src), "-S", "--cuda-device-only", "-Rpass-analysis=kernel-resource-usage", "-o",
                        str(tmp_path / "res.s"), src], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    seen = {}
    for m in re.finditer(r"Function Name: (\S+).*?VGPRs: (\d+).*?ScratchSize \[bytes/lane\]: (\d+).*?Occupancy \[waves/SIMD\]: (\d+)", r.stderr, re.S):
        seen[m.group(1)] = tuple(int(m.group(i)) for i in (2, 3, 4))
    for key, max_vgprs, min_occ in (("rtn_resident_groups", 128, 4), ("rtn_tensor_onepass", 256, 2)):
        hits = [v for k, v in seen.items() if key in k]
        assert hits, (key, list(seen))
        for vgprs, scratch, occ in hits:
            assert vgprs <= max_vgprs and scratch == 0 and occ >= min_occ, (key, vgprs, scratch, occ)
    # the per-tensor kernel's parks: the code object holds exactly the accumulation-register traffic the source writes (two parks
    # of 64 registers, each written at one place and read back at one place) and no other use of an `a` register: the compiler
    # turns to them only once the architectural registers run out (<= 128 here), and then it would take a[0], a[1], ... -- the parks
    text = (tmp_path / "res.s").read_text()
    m = re.search(r"^_ZN2oq18rtn_tensor_onepassILi8EEEvNS_12ResidentArgsE:.*?s_endpgm", text, re.S | re.M)
    assert m, "kernel body not found"
    body = m.group(0)
    writes = re.findall(r"v_accvgpr_write_b32 (\S+),", body)
    reads = re.findall(r"v_accvgpr_read_b32 \S+, (\S+)", body)
    assert len(writes) == 128 and len(reads) == 128 and len(set(writes)) == 128 and len(set(reads)) == 128, (len(writes), len(reads))
    assert all(w.startswith("a[") for w in writes) and all(r.startswith("a[") for r in reads), "an accumulation register the source did not name"
    assert "v_accvgpr_mov" not in body
    others = [ln for ln in body.splitlines() if "accvgpr" not in ln and re.search(r"[ ,]a(\[|\d)", ln.split(";")[0])]
    assert not others, others[:5]
    meta = [c for c in text.split("- .agpr_count:")[1:] if "_ZN2oq18rtn_tensor_onepassILi8EEEvNS_12ResidentArgsE\n" in c.split(".vgpr_count:")[0]]
    assert len(meta) == 1
    agpr = int(meta[0].split()[0])
    total = int(re.search(r"\.vgpr_count:\s+(\d+)", meta[0]).group(1))
    assert agpr == 128 and total <= 256, (agpr, total)      # the descriptor reserves a[0..127]; with the architectural ones: two waves per SIMD


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not available")
@pytest.mark.parametrize("source,kernels", [("rtn_mse.hip", ("mse_rows_reg_kernel",)), ("hqq.hip", ("hqq_rounds_reg_kernel",)),
                                            ("awq.hip", ("awq_group_diff_kernel",))])
def test_register_tile_search_kernels_do_not_spill(tmp_path, source, kernels):
    """(awq.hip: the quantize-residual kernel keeps a block's rows in registers from the load to the difference and, since round 5,
    through the packing of the loss product's fp16 pieces.)
    The MSE and HQQ searches hold a group's G values in registers across all candidates / rounds.  Left to itself the
    optimiser interleaves independent chunks until the G = 128 tile spills (1.1 KB of scratch per lane and 3x the time, with
    every parity test still green): the chunks are chained through opaque copies and this test watches the result."""
    from onnx_quantize_amd import _build
    src = os.path.join(ROOT, "onnx_quantize_amd", "csrc", source)
    r = subprocess.run([HIPCC, *_build.flags_for(src), "-S", "--cuda-device-only", "-Rpass-analysis=kernel-resource-usage", "-o",
                        str(tmp_path / "k.s"), src], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    seen = {}
    for m in re.finditer(r"Function Name: (\S+).*?VGPRs: (\d+).*?ScratchSize \[bytes/lane\]: (\d+).*?Occupancy \[waves/SIMD\]: (\d+)", r.stderr, re.S):
        seen[m.group(1)] = tuple(int(m.group(i)) for i in (2, 3, 4))
    for key in kernels:
        hits = {k: v for k, v in seen.items() if key in k}
        assert hits, (key, list(seen))
        for name, (vgprs, scratch, occ) in hits.items():
            assert scratch == 0 and occ >= 2, (name, vgprs, scratch, occ)
