"""Host-side check of the run-time specialisation path (gfw_jit.hip): the kernel source embedded in libgfwarp.so compiles with hiprtc
for gfx950 — no device needed — into one baked instantiation that keeps its register budget."""
import ctypes as C
import os
import sys

import pytest

from gyroflow_amd import abi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import kernel_resources as KR  # noqa: E402

C2_DEFS = "GFW_FRAME_KIND=2;GFW_FRAME_TAPS=%d;GFW_JIT_WAVES=%d;GFW_JIT_MODEL=1;GFW_JIT_T=uint16_t;GFW_JIT_N0=1;GFW_JIT_DW=2;GFW_JIT_DH=1;GFW_JIT_IL=0;GFW_JIT_RB=4;GFW_JIT_FAST1=1"


def compile_c2(tmp_path, taps, waves):
    lib = abi.load_library()
    header = open(os.path.join(ROOT, "tools", "bake_c2.h")).read()
    out = str(tmp_path / "jit.co")
    log = C.create_string_buffer(1 << 16)
    n = lib.gfw_debug_jit_compile(b"gfx950", (C2_DEFS % (taps, waves)).encode(), header.encode(), out.encode(), log, len(log))
    if n == -2:
        pytest.skip("libhiprtc.so not available")
    assert n > 0, log.value.decode(errors="replace")[-3000:]
    ks = [k for k in KR.report(out) if k[".name"] == "gfw_jit_kernel"]
    assert len(ks) == 1
    return ks[0]


def test_embedded_source_compiles_into_the_c2_instantiation(tmp_path):
    k = compile_c2(tmp_path, 2, 7)
    assert k[".vgpr_count"] <= 73 and k[".private_segment_fixed_size"] == 0, k          # seven waves per SIMD, no scratch
    assert KR.workgroups_per_cu(k) >= 7, KR.workgroups_per_cu(k)
    # the clip-invariant arguments are literals: far fewer scalar registers than the ahead-of-time kernel's 106
    assert k[".sgpr_count"] <= 100, k[".sgpr_count"]


@pytest.mark.parametrize("taps", [4, 8])
def test_embedded_source_compiles_for_the_lut_samplers(tmp_path, taps):
    k = compile_c2(tmp_path, taps, 6)                 # gfw_api.hip jit_waves: six waves per SIMD for the LUT samplers of integer planes, four tap rows in flight
    assert k[".vgpr_count"] <= 80 and k[".private_segment_fixed_size"] == 0, (k[".vgpr_count"], k[".private_segment_fixed_size"])


def test_a_broken_bake_header_is_reported_not_fatal(tmp_path):
    lib = abi.load_library()
    log = C.create_string_buffer(1 << 14)
    n = lib.gfw_debug_jit_compile(b"gfx950", (C2_DEFS % (2, 7)).encode(), b"// a bake header that defines nothing\n", b"", log, len(log))
    if n == -2:
        pytest.skip("libhiprtc.so not available")
    assert n == -1 and b"GFW_BK_" in log.value


@pytest.mark.parametrize("model,extras,waves,jit_model", [(2, 0, 8, -1), (9, 8, 8, -1), (7, 4 | 1, 8, -1), (8, 8 | 2, 8, -1), (3, 2, 8, -1)])
def test_generic_model_body_specialises_without_scratch(tmp_path, model, extras, waves, jit_model):
    """Other lens models and feature bits: the generic-model body with the model and the bits as literals (gfw_api.hip jit_for) — the ahead-of-time
    instantiation of the same body spills (tests/test_kernel_resources.py), the specialised one must not (a few spilled dwords at most under the blend of two lens solvers)."""
    lib = abi.load_library()
    header = open(os.path.join(ROOT, "tools", "bake_c2.h")).read()
    header = header.replace("#define GFW_BK_model (1)", "#define GFW_BK_model (%d)" % model).replace("#define GFW_BK_extras (0)", "#define GFW_BK_extras (%d)" % extras)
    assert "GFW_BK_model (%d)" % model in header and "GFW_BK_extras (%d)" % extras in header
    if extras & 2:
        header = header.replace("#define GFW_BK_digital (0)", "#define GFW_BK_digital (10)")
    defs = (C2_DEFS % (2, waves)).replace("GFW_JIT_MODEL=1", "GFW_JIT_MODEL=%d" % jit_model).replace("GFW_JIT_RB=4;GFW_JIT_FAST1=1", "GFW_JIT_RB=1;GFW_JIT_FAST1=0")
    out = str(tmp_path / "jit.co")
    log = C.create_string_buffer(1 << 16)
    n = lib.gfw_debug_jit_compile(b"gfx950", defs.encode(), header.encode(), out.encode(), log, len(log))
    if n == -2:
        pytest.skip("libhiprtc.so not available")
    assert n > 0, log.value.decode(errors="replace")[-3000:]
    k = [k for k in KR.report(out) if k[".name"] == "gfw_jit_kernel"][0]
    assert k[".private_segment_fixed_size"] <= (128 if extras & 8 else 0) and k[".vgpr_count"] <= (64 if waves == 8 else 80), (k[".vgpr_count"], k[".private_segment_fixed_size"])


def test_420_specialisation_keeps_eight_workgroups_per_cu(tmp_path):
    """4:2:0 (a lane owns 4 x 4 luma pixels of a tile): the first pass's LDS staging must leave room for eight workgroups per CU — it took
    37 KB per workgroup (four workgroups, 60 us per 4K NV12 frame) before the row indices went to 16 bits and the queue to a fixed size (45 us)."""
    import re
    import struct
    lib = abi.load_library()
    header = open(os.path.join(ROOT, "tools", "bake_c2.h")).read()

    def seti(name, v):
        nonlocal header
        header, n = re.subn(r"#define GFW_BK_%s \([^\n]*\)" % name, "#define GFW_BK_%s (%d)" % (name, v), header)
        assert n == 1, name

    def setf(name, v):
        nonlocal header
        bits = struct.unpack("<I", struct.pack("<f", v))[0]
        header, n = re.subn(r"#define GFW_BK_%s __builtin_bit_cast\(float, 0x[0-9a-f]+u\)" % name, "#define GFW_BK_%s __builtin_bit_cast(float, 0x%08xu)" % (name, bits), header)
        assert n == 1, name

    for k, v in (("nplanes", 2), ("ch", 1080), ("tiles_y", 68), ("pl0_src_stride", 3840), ("pl0_dst_stride", 3840), ("pl1_src_stride", 3840),
                 ("pl1_dst_stride", 3840), ("pl1_h", 1080), ("pl2_src_stride", 0), ("pl2_dst_stride", 0), ("pl2_w", 0), ("pl2_h", 0)):
        seti(k, v)
    for k, v in (("map_cy_mul", 1080.0), ("pl0_limit", 255.0), ("pl1_limit", 255.0), ("pl1_bg_0", 127.5), ("pl1_bg_1", 127.5)):
        setf(k, v)
    defs = "GFW_FRAME_KIND=1;GFW_FRAME_TAPS=2;GFW_JIT_WAVES=8;GFW_JIT_MODEL=1;GFW_JIT_T=uint8_t;GFW_JIT_N0=1;GFW_JIT_DW=2;GFW_JIT_DH=2;GFW_JIT_IL=1;GFW_JIT_RB=4;GFW_JIT_FAST1=1"
    out = str(tmp_path / "nv12.co")
    log = C.create_string_buffer(1 << 16)
    n = lib.gfw_debug_jit_compile(b"gfx950", defs.encode(), header.encode(), out.encode(), log, len(log))
    if n == -2:
        pytest.skip("libhiprtc.so not available")
    assert n > 0, log.value.decode(errors="replace")[-3000:]
    k = [k for k in KR.report(out) if k[".name"] == "gfw_jit_kernel"][0]
    assert k[".group_segment_fixed_size"] <= 20 * 1024 and k[".vgpr_count"] <= 64 and k[".private_segment_fixed_size"] == 0, k
    assert KR.workgroups_per_cu(k) >= 8, KR.workgroups_per_cu(k)


def test_a_compiler_option_travels_through_the_definition_list(tmp_path):
    """An entry of the definition list (GFW_JIT_DEFS at run time) that starts with '-' is a compiler option and comes after the library's own.  Since
    round 4 the library builds without LLVM's SLP vectoriser (46.2 against 53.5 us per C2 frame on MI355X, profiles/r04_ab_fastrow.txt): the default build
    holds next to no packed-f32 instruction, and `-fslp-vectorize` through the list brings the packing back (dozens of v_pk_*)."""
    import subprocess
    lib = abi.load_library()
    header = open(os.path.join(ROOT, "tools", "bake_c2.h")).read()
    counts = {}
    for tag, extra in (("default", ""), ("slp", ";-fslp-vectorize")):
        out = str(tmp_path / ("jit_%s.co" % tag))
        log = C.create_string_buffer(1 << 16)
        n = lib.gfw_debug_jit_compile(b"gfx950", ((C2_DEFS % (2, 8)) + extra).encode(), header.encode(), out.encode(), log, len(log))
        if n == -2:
            pytest.skip("libhiprtc.so not available")
        assert n > 0, log.value.decode(errors="replace")[-3000:]
        dis = subprocess.check_output(["/opt/rocm/lib/llvm/bin/llvm-objdump", "-d", "--mcpu=gfx950", out]).decode()
        counts[tag] = sum(1 for l in dis.splitlines() if "\tv_pk_mul_f32" in l or "\tv_pk_add_f32" in l or "\tv_pk_fma_f32" in l)
        k = [k for k in KR.report(out) if k[".name"] == "gfw_jit_kernel"][0]
        if tag == "default":
            assert k[".vgpr_count"] <= 64 and k[".private_segment_fixed_size"] == 0, k
    assert counts["default"] <= 8 and counts["slp"] > 50, counts


def test_the_checksum_build_takes_its_sums_without_a_private_segment(tmp_path):
    """gfw_set_frame_checksums: the checksum build of C2's kernel (GFW_BK_checksum = 1) as jit_waves budgets it — seven waves per SIMD.  What the option costs is
    registers, not instructions: builds whose fold left lane-derived values in scratch across the kernel ran 4-10 % slower on the MI355X than their scratch-free twins
    (same instruction count, all waves resident: profiles/r05_c5_checksum.txt), so the shipped form must have none; its LDS grows by the 256 lane slots and the four wave words."""
    lib = abi.load_library()
    plain = open(os.path.join(ROOT, "tools", "bake_c2.h")).read()
    assert "#define GFW_BK_checksum (0)" in plain
    got = {}
    for name, header in (("plain", plain), ("checksum", plain.replace("#define GFW_BK_checksum (0)", "#define GFW_BK_checksum (1)"))):
        out = str(tmp_path / (name + ".co"))
        log = C.create_string_buffer(1 << 16)
        n = lib.gfw_debug_jit_compile(b"gfx950", (C2_DEFS % (2, 7)).encode(), header.encode(), out.encode(), log, len(log))
        if n == -2:
            pytest.skip("libhiprtc.so not available")
        assert n > 0, log.value.decode(errors="replace")[-3000:]
        got[name] = [k for k in KR.report(out) if k[".name"] == "gfw_jit_kernel"][0]
    k, kp = got["checksum"], got["plain"]
    assert k[".private_segment_fixed_size"] == 0 and k[".vgpr_count"] <= 72, (k[".vgpr_count"], k[".private_segment_fixed_size"])
    assert KR.workgroups_per_cu(k) >= 7, KR.workgroups_per_cu(k)
    assert k[".group_segment_fixed_size"] - kp[".group_segment_fixed_size"] in range(2048, 2048 + 128), (k[".group_segment_fixed_size"], kp[".group_segment_fixed_size"])


@pytest.mark.parametrize("model", ["gopro", "sony", "generic_polynomial", "opencv_standard", "poly3", "poly5", "ptlens", "insta360"])
def test_every_lens_models_clip_specialises_without_a_private_segment(tmp_path, model):
    """Round 6: the twelve-coefficient polynomial's certified build came out with 2.3 KB of scratch per lane — the kernel's whole argument block, copied there by every
    lane because the body TESTED its pointer (`clip ? clip->n_frames : 1`; a comparison is a use the optimiser cannot forward to the argument segment) — ran 0.43 ms
    per launch and, the copy sitting at private offset 0 where the compare reads "null", wrote only the first frame of a clip launch (profiles/r06_radial_closed_form.txt).
    Whatever a lens model's clip compiles to reads its arguments where they are (54 builds — 9 models x 3 formats x bilinear / Lanczos4 — swept by hand: at most
    20 bytes, a few spilled dwords of the GoPro solver; this keeps the C2 geometry of each model under test)."""
    import build_jit_cache as B
    from gyroflow_amd import synthetic as S
    import bench
    lib = abi.load_library()
    lens = dict(S.gopro_style_lens(3840, 2160))
    k = bench.LENS_MODEL_K[model]
    lens["model"], lens["k"] = model, k + [0.0] * (12 - len(k))
    fr = S.SyntheticFrame("YUV422P16LE", 3840, 2160, seed=0x9F10, timestamp_ms=1000.0, lens=lens, readout_ms=16.0, pixels=False)
    defs, header, _ = B.key_of(lib, fr)
    served = model in ("gopro", "sony", "generic_polynomial")                            # the certified first pass over r (gfw_api_certificate.inc: p1_model_radial_served)
    assert (b"GFW_JIT_FAST1=1" in defs and b"#define GFW_P1_RFORM (1)" in header) == served, defs
    out = str(tmp_path / "radial.co")
    log = C.create_string_buffer(1 << 16)
    n = lib.gfw_debug_jit_compile(b"gfx950", defs, header, out.encode(), log, len(log))
    if n == -2:
        pytest.skip("libhiprtc.so not available")
    assert n > 0, log.value.decode(errors="replace")[-3000:]
    kr = [x for x in KR.report(out) if x[".name"] == "gfw_jit_kernel"][0]
    # (the GoPro solver's ten Newton steps spill three dwords at eight waves — the argument block is 2224 bytes)
    assert kr[".private_segment_fixed_size"] <= 64 and kr[".vgpr_count"] <= 64, (kr[".vgpr_count"], kr[".private_segment_fixed_size"])
