"""Toolchain guard (no GPU needed: hipcc cross-compiles): every MFMA kernel is compiled to gfx950 ISA and
scanned for VALU reads of an MFMA result with too few wait states, including across branches — the defect
ROCm 7.2's hipcc showed on `if (relu)` behind an accumulator (see sb.h relu_opt)."""
import glob
import importlib.util
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "esa-pose-estimation_amd", "csrc")
HIPCC = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"

spec = importlib.util.spec_from_file_location("isa_hazard_check", os.path.join(ROOT, "tools", "isa_hazard_check.py"))
chk = importlib.util.module_from_spec(spec)
spec.loader.exec_module(chk)

MFMA_SOURCES = sorted(f for f in glob.glob(os.path.join(CSRC, "*.hip")) if "mfma_f32" in open(f).read())
ALL_SOURCES = sorted(glob.glob(os.path.join(CSRC, "*.hip")))


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not installed")
@pytest.mark.parametrize("src", MFMA_SOURCES, ids=[os.path.basename(s) for s in MFMA_SOURCES])
def test_no_unpadded_mfma_result_reads(src, tmp_path):
    out = tmp_path / (os.path.basename(src) + ".s")
    r = subprocess.run([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-I" + os.path.join(ROOT, "include"),
                        "-S", "--cuda-device-only", "-o", str(out), src], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    found = chk.scan(str(out))
    assert not found, found[:5]


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not installed")
@pytest.mark.parametrize("src", ALL_SOURCES, ids=[os.path.basename(s) for s in ALL_SOURCES])
def test_no_store_data_overwritten_behind_a_wide_store(src, tmp_path):
    """gfx950 reads the data registers of 12/16-byte stores late (lanes 12..15 of each row of 16): nothing may write
    them for 2 wait states.  hipcc pads global stores and buffer stores with an immediate soffset, NOT buffer stores
    with an SGPR soffset (tools/ubench/store_data_war.hip) — the source of the 'columns 12..15' corruptions of round 1."""
    out = tmp_path / (os.path.basename(src) + ".s")
    r = subprocess.run([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-I" + os.path.join(ROOT, "include"),
                        "-S", "--cuda-device-only", "-o", str(out), src], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    found = chk.scan_store_data(str(out))
    assert not found, found[:5]


def test_store_scanner_sees_a_known_bad_sequence(tmp_path):
    bad = tmp_path / "bad.s"
    bad.write_text("\n".join([
        "\tbuffer_store_dwordx4 v[142:145], v146, s[72:75], s58 offen",
        "\tv_max_i32_e32 v142, 0, v134",
        "\tglobal_store_dwordx4 v[10:11], v[20:23], off",
        "\ts_nop 0",
        "\tv_mov_b32_e32 v23, 0",
        "\tglobal_store_dwordx4 v[10:11], v[30:33], off",
        "\ts_nop 1",
        "\tv_mov_b32_e32 v33, 0",
        "\ts_endpgm", ""]))
    found = chk.scan_store_data(str(bad))
    assert [f[1] for f in found] == [0, 1]


def test_scanner_sees_a_known_bad_sequence(tmp_path):
    bad = tmp_path / "bad.s"
    bad.write_text("\n".join([
        "\tv_mfma_f32_16x16x32_bf16 v[54:57], v[54:57], v[30:33], v[50:53]",
        "\ts_cbranch_vccz .LBB2_21",
        "\ts_nop 6",
        "\tv_max_i32_e32 v54, 0, v54",
        ".LBB2_21:",
        "\tv_cvt_pk_bf16_f32 v52, v54, v55",
        "\ts_endpgm", ""]))
    found = chk.scan(str(bad))
    assert len(found) == 1 and found[0][1] < 7
