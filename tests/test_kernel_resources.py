"""Register / LDS / scratch budget of the built kernels, read from the code objects inside libgfwarp.so (no GPU needed).

The fused kernel's speed depends on how many workgroups a CU admits (profiles/r02_scheduling_experiments.md): a macro or
compiler change that pushes the hot instantiations past a register step, or into scratch, must fail here and not show up as an
unexplained slowdown on the GPU box."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import kernel_resources as KR          # noqa: E402

LIB = os.path.join(ROOT, "gyroflow_amd", "libgfwarp.so")


@pytest.fixture(scope="module")
def kernels():
    if not os.path.exists(LIB):
        pytest.skip("libgfwarp.so not built")
    ks = {k[".name"]: k for k in KR.report(LIB)}
    assert len(ks) > 100, "code objects of libgfwarp.so not found"
    return ks


def fused(kernels, model, t, n0, taps, dw, dh, il, rb, fast1, audit=False):
    """gfw_yuv_kernel<MODEL, T, N0, I, DW, DH, INTERLEAVED_UV, RB, FAST1, AUDIT> by its mangled template arguments."""
    m = "Lin1E" if model < 0 else "Li%dE" % model
    key = "gfw_yuv_kernelI%s%sLi%dELi%dELi%dELi%dELb%dELi%dELb%dELb%dEE" % (m, t, n0, taps, dw, dh, il, rb, fast1, audit)
    hits = [k for n, k in kernels.items() if key in n]
    assert len(hits) == 1, (key, len(hits))
    return hits[0]


@pytest.mark.parametrize("name,args", [
    ("C2 u16 4:2:2 planar, bilinear, certified first pass", (1, "t", 1, 2, 2, 1, 0, 4, 1)),
    ("C1 u8 NV12, bilinear, certified first pass", (1, "h", 1, 2, 2, 2, 1, 4, 1)),
    ("C1 u8 NV12, bilinear, one matrix", (1, "h", 1, 2, 2, 2, 1, 1, 0)),
    ("C4 RGBAf packed, bilinear", (1, "f", 4, 2, 1, 1, 0, 4, 1)),
    ("C4 planar f32, bilinear", (1, "f", 1, 2, 1, 1, 0, 4, 1)),
    ("u16 4:2:2 Lanczos4", (1, "t", 1, 8, 2, 1, 0, 4, 1)),
    ("u16 4:2:2 bicubic", (1, "t", 1, 4, 2, 1, 0, 4, 1)),
])
def test_specialised_fisheye_instantiations_keep_six_workgroups_per_cu(kernels, name, args):
    k = fused(kernels, *args)
    assert k[".vgpr_count"] <= 80, (name, k[".vgpr_count"])                       # 512 / 80 = 6 waves per SIMD
    assert k[".sgpr_count"] <= 112, (name, k[".sgpr_count"])                      # floor(800 / (112 + 16)) = 6 workgroups per CU
    assert k[".private_segment_fixed_size"] <= 128, (name, k[".private_segment_fixed_size"])   # the rolled edge path's arrays, no spills
    dh_lds_limited = args[5] == 2                                                 # 4:2:0 tiles carry twice the rows: LDS admits 4
    assert KR.workgroups_per_cu(k) >= (4 if dh_lds_limited else 6), (name, KR.workgroups_per_cu(k))


def test_generic_model_instantiations_keep_their_register_budget(kernels):
    """-1: every other lens model / digital lens / refraction / IBIS / blend; -2: the same plus background mode 3 and the Sony mesh."""
    for tag in ("gfw_yuv_kernelILin1E", "gfw_yuv_kernelILin2E"):
        gen = [k for n, k in kernels.items() if tag in n]
        assert len(gen) >= 30, (tag, len(gen))
        for k in gen:
            bilinear = "ELi2ELi" in k[".name"].split(tag)[1][:24]
            assert KR.workgroups_per_cu(k) >= 3, (k[".name"], k[".vgpr_count"], k[".group_segment_fixed_size"])
            # six waves per SIMD (measured faster on a digital-lens clip) cost the bilinear ones up to ~1.5 KB of scratch per lane
            assert k[".private_segment_fixed_size"] <= (2048 if bilinear or tag.endswith("n2E") else 768), (k[".name"], k[".private_segment_fixed_size"])
        # only the instantiations the dispatcher can reach are built: the certified first pass exists for the fisheye model alone
        assert not [k for k in gen if "ELi4ELb1ELb" in k[".name"]]


def test_no_kernel_is_left_with_one_wave_per_simd(kernels):
    for n, k in kernels.items():
        if "gfw_" in n:
            assert KR.waves_per_simd(k[".vgpr_count"]) >= 2, (n, k[".vgpr_count"])
