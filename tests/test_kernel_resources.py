"""Register / LDS budgets of the two pixel kernels, checked offline from hipcc's own metadata (no GPU needed).  E1's three resident workgroups
per CU rest on 80 VGPRs (6 waves per SIMD x 80 <= 512) and 3 x LDS <= 160 KB; W1's three on its LDS footprint.  The numbers moved with compiler
flags and harmless-looking source changes during round 4 (the SLP vectorizer: 146; a refactor into lambdas: 81), so the budget is a test."""
import glob
import os
import re
import shutil
import subprocess
import tempfile

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = "/opt/rocm/bin/hipcc"


def _census(src):
    flags = None
    for ln in open(os.path.join(ROOT, "visiondepth3d_amd", "csrc", "Makefile")):
        if ln.startswith("FLAGS"):
            flags = ln.split("=", 1)[1].replace("$(ARCH)", "gfx950").split()
    assert flags and "-fno-slp-vectorize" in flags
    d = tempfile.mkdtemp()
    try:
        subprocess.run([HIPCC, *[f for f in flags if f != "-Wall"], "-I" + os.path.join(ROOT, "visiondepth3d_amd", "csrc"), "-c",
                        os.path.join(ROOT, "visiondepth3d_amd", "csrc", src), "-o", "x.o", "--save-temps"], cwd=d, check=True,
                       stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=600)
        asm = open(glob.glob(os.path.join(d, "*gfx950.s"))[0]).read()
    finally:
        shutil.rmtree(d, ignore_errors=True)
    out = {}
    for m in re.finditer(r"\.group_segment_fixed_size: (\d+).*?\.name:\s+(\S+).*?\.vgpr_count:\s+(\d+).*?\.vgpr_spill_count: (\d+)", asm, re.S):
        out[m.group(2)] = dict(lds=int(m.group(1)), vgpr=int(m.group(3)), spill=int(m.group(4)))
    return out


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="needs hipcc")
def test_e1_fits_three_workgroups_per_cu():
    k = _census("vd3d_finish.hip")
    e1 = next(v for n, v in k.items() if n.startswith("_Z14k_finish_fusedILb1ELi26E"))
    assert e1["spill"] == 0 and e1["vgpr"] <= 80, e1          # 8 waves per workgroup = 2 per SIMD; three workgroups = 6 waves x 80 VGPRs <= 512
    assert 3 * e1["lds"] <= 160 * 1024, e1


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="needs hipcc")
def test_w1_and_mask_kernel_budgets():
    k = _census("vd3d_warp.hip")
    for n, v in k.items():
        assert v["spill"] == 0, (n, v)
    w1 = next(v for n, v in k.items() if n.startswith("_Z12k_warp_fusedILb1ELb1ELi32ELb1E"))
    assert w1["vgpr"] <= 80 and w1["lds"] % 16 == 0, w1       # static LDS in front of the dynamic array keeps it 16-byte aligned (ds_read_b128)
    # round 6: the no-feather kernel that computes the shift plane of its own tile (pow / sigmoid / sqrt chains in its prologue) and the feathered kernel whose
    # S loads once moved behind the E2 tile load (92 VGPRs, two workgroups per CU, 115 us instead of 87): three resident workgroups = 6 waves per SIMD = 85 registers
    fold = next(v for n, v in k.items() if n.startswith("_Z12k_warp_fusedILb1ELb0ELi32ELb0ELb1E"))
    fold1 = next(v for n, v in k.items() if n.startswith("_Z12k_warp_fusedILb0ELb0ELi32ELb0ELb1E"))
    assert fold["vgpr"] <= 84 and fold1["vgpr"] <= 84, (fold, fold1)
    e2w = next(v for n, v in k.items() if n.startswith("_Z5k_e2w"))
    assert e2w["vgpr"] <= 64 and e2w["lds"] <= 20 * 1024, e2w


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="needs hipcc")
def test_scalar_tail_instantiations_leave_the_video_size_kernels_alone():
    """Round 5: `k_shift` and `k_chain_shape` exist with and without ATen's scalar-tail arithmetic (fp64 `pow`, glibc's `expf` in double).  Every video-sized plane launches
    the <false> instantiation, which must keep the registers it had before the mode existed (20 / 52); the <true> ones may be as fat as they like but must not spill."""
    k = _census("vd3d_planes.hip")
    sh_plain = next(v for n, v in k.items() if n.startswith("_Z7k_shiftILb0E"))
    sh_tails = next(v for n, v in k.items() if n.startswith("_Z7k_shiftILb1E"))
    assert sh_plain["vgpr"] <= 20 and sh_plain["spill"] == 0, sh_plain
    assert sh_tails["spill"] == 0, sh_tails
    k = _census("vd3d_select.hip")
    cs_plain = next(v for n, v in k.items() if n.startswith("_Z13k_chain_shapeILb0E"))
    cs_tails = next(v for n, v in k.items() if n.startswith("_Z13k_chain_shapeILb1E"))
    assert cs_plain["vgpr"] <= 52 and cs_plain["spill"] == 0, cs_plain
    assert cs_tails["spill"] == 0 and cs_tails["vgpr"] <= 128, cs_tails       # 1024 threads per workgroup: 4 waves per SIMD x 128 VGPRs


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="needs hipcc")
def test_split_bf16_kernels_fit_two_waves_per_simd():
    """Round 6: k_gemm_bf16x3 and k_attn_bf16x3 are 512-thread workgroups (two waves per SIMD: 256 registers each at most) that keep 128 / 64 accumulator registers
    plus their fragments live across a software-pipelined loop -- one spilled register inside that loop would cost more than any schedule could win back."""
    k = _census("vd3d_gemm.hip")
    g = next(v for n, v in k.items() if n.startswith("_Z13k_gemm_bf16x3ILi0E"))
    assert g["spill"] == 0 and g["vgpr"] <= 256, g
    k = _census("vd3d_attn.hip")
    a = next(v for n, v in k.items() if n.startswith("_Z13k_attn_bf16x3"))
    assert a["spill"] == 0 and a["vgpr"] <= 256, a
    k = _census("vd3d_conv2.hip")
    convs = [v for n, v in k.items() if n.startswith("_Z12k_conv3x3_x2ILi")]
    assert len(convs) == 3, sorted(k)                                   # 32, 64 and 128 output channels
    for c in convs:
        assert c["spill"] == 0 and c["vgpr"] <= 256 and c["lds"] <= 160 * 1024, c
