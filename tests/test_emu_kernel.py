"""The fused kernel's SOURCE on the host cores (tests/_emu.py: the text embedded for hiprtc, compiled as C++ for x86-64 and interpreted lane by
lane) against the fixture the reference's own kernel wrote and against the oracle — the CPU tier's check of the product's device code itself:
its arithmetic (lean divide / sqrt / table atanf, the exact projection, the taps), the persistent tile walk and clip launch, the certified
first pass with its LDS queue, chroma sites from luma coordinates, the generic-model bodies with every lens model and feature bit as a literal.

What the interpreter cannot stand for is the hardware itself (v_rcp_f32 / v_sqrt_f32 are exact here and refined from there, wave votes answer
per lane, occupancy and timing do not exist): the GPU tier runs the same checks on the MI355X.  Nothing here is a product path."""
import json
import os
import sys
import zlib

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))
import make_ref_golden as G  # noqa: E402

from gyroflow_amd import abi, synthetic as S  # noqa: E402
import _emu  # noqa: E402
import _oracle as O  # noqa: E402

GOLD = json.load(open(os.path.join(HERE, "golden", "ref_golden.json")))
FUSED = sorted(n for n in G.CASES if _emu.fused_eligible(G.build(dict(G.CASES[n], w=64, h=32, out=None) if G.CASES[n]["w"] > 2000 else G.CASES[n])))


def test_the_interpreter_covers_the_fused_cases_of_the_fixture():
    assert len(FUSED) >= 38 and "c2_yuv422p16_3840x2160_rs" in FUSED, FUSED


@pytest.mark.parametrize("name", FUSED)
def test_kernel_source_reproduces_the_reference_kernels_output(name):
    """gfw_frame.hip (host-interpreted) == tests/golden/ref_golden.json, i.e. what the reference's opencl_undistort.cl wrote — C2 at 3840x2160 included"""
    fr = G.build(G.CASES[name])
    planes = _emu.run_frame(fr)
    assert [zlib.crc32(p.tobytes()) for p in planes] == GOLD[name]["planes"]


AOT_CASES = ["c2_yuv422p16_480x270_rs", "c1_nv12_1920x1080_constquat", "p010_lanczos4_640x360", "yuv420p_bilinear_642x362", "yuv444p16_bicubic_640x360", "rgba_lanczos4_640x360",
             "gbrapf32_bicubic_640x360", "nv12_horizontal_rs", "yuv422p16_mirror", "gopro_640x360", "superview_nv12_lca06_640x360", "hyperview_lca06_640x360", "ibis_terms_640x360"]


@pytest.mark.parametrize("name", AOT_CASES)
def test_ahead_of_time_form_reproduces_the_reference_kernels_output(name):
    """the same body as the ahead-of-time kernels run it: every clip-invariant field read from the argument block (GFW_BAKE = 0), the generic-model
    instantiation for any feature bit"""
    fr = G.build(G.CASES[name])
    assert [zlib.crc32(p.tobytes()) for p in _emu.run_frame(fr, baked=False)] == GOLD[name]["planes"]


def same_as_oracle(fr, mesh=None):
    for i, (a, b) in enumerate(zip(O.run_frame(fr) if mesh is None else [], _emu.run_frame(fr, mesh, baked=False) if mesh is None else [])):
        assert np.array_equal(a, b), "ahead-of-time form, plane %d: %d bytes differ" % (i, int(np.count_nonzero(a != b)))
    got = _emu.run_frame(fr, mesh)
    for i, pl in enumerate(fr.planes):
        ref = pl["dst"].copy()
        assert O.undistort_image(pl["src"], pl["size"], ref, pl["out_size"], pl["params"], pl["pixel_type"], fr.model, fr.digital, fr.matrices, mesh=mesh) == 1
        assert np.array_equal(ref, got[i]), "plane %d: %d bytes differ" % (i, int(np.count_nonzero(ref != got[i])))


@pytest.mark.parametrize("fmt,interp", [("YUV422P16LE", 2), ("NV12", 2), ("P010LE", 8), ("YUV420P", 4), ("RGBA", 2), ("RGBAF32", 2)])
@pytest.mark.parametrize("fov,hrs", [(1.7, False), (3.0, True)])
def test_zoomed_out_frames_against_the_oracle(fmt, interp, fov, hrs):
    """out-of-frame pixels, edge taps, negative coordinates (where the reference's GPU twin deviates from its CPU path and no fixture can exist)"""
    same_as_oracle(S.SyntheticFrame(fmt, 322, 186, seed=0x51, fov=fov, horizontal_rs=hrs, interpolation=interp, readout_ms=25.0, background_rgba=(0.3, 0.6, 0.9, 1.0)))


@pytest.mark.parametrize("model,k", [("opencv_standard", [0.12, -0.05, 0.001, 0.002, 0.01, 0.02, -0.01, 0.001, 0.0005, -0.0002, 0.0003, 0.0001]), ("poly3", [0.06]),
                                     ("poly5", [0.08, -0.02]), ("insta360", [0.05, -0.01, 0.002, 0.001, -0.001, 0.6]), ("gopro", [0.0, 1.0, 0.01, -0.12, 0.02, 0.01, -0.004])])
@pytest.mark.parametrize("lca", [1.0, 0.45])
def test_lens_models_and_blend_against_the_oracle(model, k, lca):
    lens = S.gopro_style_lens(320, 192)
    lens["model"], lens["k"] = model, k + [0.0] * (12 - len(k))
    if model == "gopro":
        lens["r_limit"] = 2.5
    same_as_oracle(S.SyntheticFrame("YUV422P16LE", 320, 192, seed=31, lens=lens, fov=1.2, base_overrides={"lens_correction_amount": lca}))


def test_feature_bits_against_the_oracle():
    w, h = 320, 192
    lens = S.gopro_style_lens(w, h)
    lens["r_limit"] = 0.9
    same_as_oracle(S.SyntheticFrame("YUV422P16LE", w, h, seed=41, lens=lens, fov=1.6))                                          # r-limit
    same_as_oracle(S.SyntheticFrame("YUV422P16LE", w, h, seed=82, base_overrides={"light_refraction_coefficient": 1.33}, flags=2048))   # refraction, NaN coordinates
    same_as_oracle(S.SyntheticFrame("NV12", w, h, seed=78, fov=1.3, base_overrides={"background_mode": 3, "background_margin": 0.1, "background_margin_feather": 0.05}))
    same_as_oracle(S.SyntheticFrame("YUV422P16LE", w, h, seed=78, fov=2.2, base_overrides={"background_mode": 2}))
    lens = S.gopro_style_lens(w, h)
    lens["digital"] = "gopro_hyperview"
    same_as_oracle(S.SyntheticFrame("YUV422P16LE", w, h, seed=37, lens=lens, fov=1.3))                                          # a diverging digital-lens inverse


@pytest.mark.parametrize("with_mesh,with_fpd,inverted", [(True, True, False), (True, True, True), (False, True, False)])
def test_sony_mesh_against_the_oracle(with_mesh, with_fpd, inverted):
    from test_ref_opencl_host import _mesh_block
    w, h = 192, 128
    fr = S.SyntheticFrame("NV12", w, h, seed=47, fov=1.1, flags=abi.FLAG_FRAMEBUFFER_INVERTED if inverted else 0)
    same_as_oracle(fr, _mesh_block(w, h, with_fpd, with_mesh))


def test_large_rotation_fills_the_first_pass_queue():
    """a fast pan with a long readout: the certified first pass rejects many more pixels than usual; they go through the LDS queue and the exact pass"""
    fr = S.SyntheticFrame("YUV422P16LE", 640, 360, seed=0x77, fov=1.1, readout_ms=60.0)
    same_as_oracle(fr)
    k = fr.planes[0]["params"].k
    assert _emu.p1_table(fr.planes[0]["params"], fr.matrices, fr.planes[0]["params"].matrix_count) is not None and any(k[i] != 0.0 for i in range(4))


def test_clip_launch_of_five_frames():
    """gfw_undistort_clip's launch shape: the (frame, tile) pairs of several frames dealt to the persistent workgroups, every XCD slot taking another
    region of each frame; per-frame plane pointers and matrix tables from the argument block"""
    frames = [S.SyntheticFrame("YUV422P16LE", 384, 208, seed=0x9F10 + j, timestamp_ms=1000.0 + 33.3 * j) for j in range(5)]
    outs = _emu.run_frames(frames)
    for fr, got in zip(frames, outs):
        ref = O.run_frame(fr)
        assert all(np.array_equal(a, b) for a, b in zip(ref, got))


@pytest.mark.parametrize("fmt,kw", [
    ("YUV422P16LE", dict(fov=1.4, base_overrides={"background_mode": 3, "background_margin": 0.1, "background_margin_feather": 0.1})),
    ("YUVA444P10LE", dict(fov=1.3, interpolation=4, base_overrides={"background_mode": 3, "background_margin": 0.13, "background_margin_feather": 0.24})),
    ("YUV420P", dict(fov=1.2, base_overrides={"lens_correction_amount": 0.5})),
    ("P010", dict(fov=1.5, base_overrides={"background_mode": 3, "background_margin": 0.05, "background_margin_feather": 0.05, "light_refraction_coefficient": 0.9}, flags=2048)),
])
def test_clip_launch_of_the_feature_bodies(fmt, kw):
    """The interpreter's random sweep found it (seed 89 of tests/test_gpu_fuzz.py's generator, then a three-frame launch): background mode 3 on planar
    chroma took the chroma planes' pointers from the argument block — frame 0's — for every frame of a clip launch.  Fixed in gfw_frame.hip (the named
    plane objects carry the current frame's pointers); the GPU twin of this test is tests/test_gpu_jit.py::test_clip_launches_of_the_generic_model_bodies."""
    frames = [S.SyntheticFrame(fmt, 192, 112, seed=100 + j, timestamp_ms=1000.0 + 33.3 * j, **kw) for j in range(3)]
    for j, (fr, got) in enumerate(zip(frames, _emu.run_frames(frames))):
        for p, (a, b) in enumerate(zip(O.run_frame(fr), got)):
            assert np.array_equal(a, b), "frame %d plane %d: %d bytes differ" % (j, p, int(np.count_nonzero(a != b)))
    for p, (a, b) in enumerate(zip(O.run_frame(frames[0]), _emu.run_frame(frames[0], baked=False))):          # and the ahead-of-time form of the same body
        assert np.array_equal(a, b), "ahead-of-time form, plane %d: %d bytes differ" % (p, int(np.count_nonzero(a != b)))


# ---- the complete per-plane operator (gfw_plane_kernel.h), which serves whatever the fused kernel does not ------------------------------

PER_PLANE = sorted(n for n in G.CASES if n not in FUSED) + ["yuv422p16_stretch", "yuv422p16_fill_background", "input_rotation_90_nv12", "input_rotation_180_640x360", "c2_yuv422p16_480x270_rs", "c1_nv12_1920x1080_constquat", "p010_lanczos4_640x360", "yuv420p_bilinear_642x362",
                                                            "rgba64_bicubic_640x360", "gbrapf32_bicubic_640x360", "yuv422p16_mirror", "gopro_640x360", "hyperview_lca06_640x360",
                                                            "ibis_terms_640x360", "rgbaf16_bilinear_640x360"]


def test_per_plane_cases_cover_what_the_fused_kernel_leaves():
    assert {"yuv422p16_stretch", "yuv422p16_fill_background", "input_rotation_90_nv12", "input_rotation_180_640x360", "rgbaf16_bilinear_640x360"} <= set(FUSED)     # the fused kernel's since round 4 / 5 (and still the per-plane kernel's when asked)
    assert {"yuv422p16_stretch", "yuv422p16_fill_background", "input_rotation_90_nv12", "input_rotation_180_640x360", "rgbaf16_bilinear_640x360"} <= set(PER_PLANE)


@pytest.mark.parametrize("name", PER_PLANE)
def test_per_plane_kernel_source_reproduces_the_reference_kernels_output(name):
    fr = G.build(G.CASES[name])
    assert [zlib.crc32(p.tobytes()) for p in _emu.run_frame_per_plane(fr)] == GOLD[name]["planes"]


def per_plane_same_as_oracle(fr, mesh=None):
    for i, pl in enumerate(fr.planes):
        ref = pl["dst"].copy()
        assert O.undistort_image(pl["src"], pl["size"], ref, pl["out_size"], pl["params"], pl["pixel_type"], fr.model, fr.digital, fr.matrices, mesh=mesh) == 1
        got = _emu.run_plane(fr, i, mesh)
        assert np.array_equal(ref, got), "plane %d: %d bytes differ" % (i, int(np.count_nonzero(ref != got)))


@pytest.mark.parametrize("interp", [10, 11, 12, 13])
def test_per_plane_ewa_against_the_oracle(interp):
    per_plane_same_as_oracle(S.SyntheticFrame("YUV422P16LE", 160, 96, seed=5, fov=1.3, interpolation=interp))


@pytest.mark.parametrize("fmt,interp,kw", [
    ("YUV422P16LE", 10, dict(fov=1.6, background_rgba=(0.9, 0.2, 0.4, 1.0))),
    ("YUV420P", 12, dict(fov=1.4, background_rgba=(0.1, 0.8, 0.3, 1.0), base_overrides={"background_mode": 1})),
    ("YUV444P16LE", 11, dict(fov=1.3, background_rgba=(0.3, 0.6, 0.9, 1.0), base_overrides={"background_mode": 3, "background_margin": 0.1, "background_margin_feather": 0.12})),
    ("YUV420P", 13, dict(fov=1.2, flags=abi.FLAG_FIX_COLOR_RANGE, limited_range=True)),
])
def test_ewa_on_paired_chroma_planes_against_the_oracle(fmt, interp, kw):
    """gfw_plane_kernel<.., DUAL> (round 6): U and V of a planar frame through ONE launch — one set of coordinates, jacobians and tap weights, two sums, each
    plane's own background — must write what the oracle writes for each plane on its own (cpu_undistort.rs:331-369 per plane)."""
    fr = S.SyntheticFrame(fmt, 160, 96, seed=15 + interp, interpolation=interp, **kw)
    assert len(fr.planes) >= 3 and fr.planes[1]["pixel_type"] == fr.planes[2]["pixel_type"]
    got = _emu.run_plane_pair(fr, 1)
    for k, i in enumerate((1, 2)):
        pl = fr.planes[i]
        ref = pl["dst"].copy()
        assert O.undistort_image(pl["src"], pl["size"], ref, pl["out_size"], pl["params"], pl["pixel_type"], fr.model, fr.digital, fr.matrices) == 1
        assert np.array_equal(ref, got[k]), "plane %d: %d bytes differ" % (i, int(np.count_nonzero(ref != got[k])))
    assert not np.array_equal(got[0], got[1])


@pytest.mark.parametrize("fmt", ["RGB24", "RGB48BE", "AYUV64LE", "RGBAF16"])
def test_per_plane_pixel_types_against_the_oracle(fmt):
    per_plane_same_as_oracle(S.SyntheticFrame(fmt, 200, 120, seed=6, fov=1.5, background_rgba=(0.2, 0.4, 0.6, 0.8)))


def test_per_plane_flags_rotation_and_mesh_against_the_oracle():
    from test_ref_opencl_host import _mesh_block
    w, h = 200, 120
    per_plane_same_as_oracle(S.SyntheticFrame("NV12", w, h, seed=84, flags=abi.FLAG_FIX_COLOR_RANGE, limited_range=True))       # the CPU path's colour-range fix
    fr = S.SyntheticFrame("YUV422P16LE", w, h, seed=43, fov=1.3)
    for pl in fr.planes:
        pl["params"].input_rotation = 17.5
    per_plane_same_as_oracle(fr)
    per_plane_same_as_oracle(S.SyntheticFrame("YUV422P16LE", w, h, seed=83, fov=1.4, base_overrides={"input_vertical_stretch": 1.1, "input_horizontal_stretch": 0.9}))
    per_plane_same_as_oracle(S.SyntheticFrame("NV12", w, h, seed=47, fov=1.1), _mesh_block(w, h, True, True))
    lens = S.gopro_style_lens(w, h)
    lens["model"], lens["k"], lens["digital"] = "sony", [1.0, 0.01, -0.05, 0.02, 0.003, -0.001] + [0.0] * 6, "gopro_superview"
    per_plane_same_as_oracle(S.SyntheticFrame("P010LE", w, h, seed=48, lens=lens, fov=1.2, base_overrides={"lens_correction_amount": 0.7, "background_mode": 3,
                                                                                                       "background_margin": 0.1, "background_margin_feather": 0.05}))


@pytest.mark.parametrize("grid", [16, 24, 256, 2048])
def test_tile_walk_under_other_launch_sizes(grid):
    """the library launches num_cus x waves persistent workgroups capped at the tile count; the (frame, sub-band, tile) walk must cover every tile once for any multiple of 8"""
    for fmt, w, h, n in (("YUV422P16LE", 384, 208, 5), ("NV12", 322, 186, 3), ("RGBA", 200, 120, 16)):
        frames = [S.SyntheticFrame(fmt, w, h, seed=0x9F10 + j, timestamp_ms=1000.0 + 33.3 * j) for j in range(n)]
        for fr, got in zip(frames, _emu.run_frames(frames, grid=grid)):
            assert all(np.array_equal(a, b) for a, b in zip(O.run_frame(fr), got)), (fmt, grid)


@pytest.mark.parametrize("name", ["c2_yuv422p16_480x270_rs", "yuv420p10_320x192_rs", "nv12_horizontal_rs", "yuv422p16_fov05_hrs", "c1_nv12_1920x1080_constquat", "superview_640x360"])
def test_the_general_side_of_every_wave_vote(name):
    """`__all(x < 0.4375)` (atanf without its reduction), `__any(tiny)` (chroma coordinate as half the luma one): the interpreter normally answers a vote with
    the lane's own predicate, so a lane only ever runs the side it qualifies for; here every vote is answered as if another lane had failed it, and all pixels
    take the general side — which must produce the same bits (on the device a single lane decides for its 63 neighbours)."""
    fr = G.build(G.CASES[name])
    assert [zlib.crc32(p.tobytes()) for p in _emu.run_frames([fr], votes=1)[0]] == GOLD[name]["planes"]


@pytest.mark.parametrize("name", ["c2_yuv422p16_480x270_rs", "nv12_horizontal_rs", "yuv422p16_fov05_hrs", "c4_rgbaf32_crop_1280x720", "sony_640x360"])
def test_lean_primitives_do_not_depend_on_which_one_ulp_answer_the_hardware_gives(name):
    """gfw_fastmath.h refines v_rcp_f32 / v_sqrt_f32 (1-ulp approximations) into correctly rounded quotients and roots.  The interpreter's stand-ins return
    the correctly rounded value; moved by one ulp either way the frame must not change — moved by two it must (the refinement's reach, and the proof that
    this run can see a wrong primitive)."""
    fr = G.build(G.CASES[name])
    for ulp in (1, -1):
        assert [zlib.crc32(p.tobytes()) for p in _emu.run_frames([fr], hw_ulp=ulp)[0]] == GOLD[name]["planes"], ulp
    if name == "c2_yuv422p16_480x270_rs":
        assert [zlib.crc32(p.tobytes()) for p in _emu.run_frames([fr], hw_ulp=2)[0]] != GOLD[name]["planes"]


@pytest.mark.parametrize("trial", range(6))
def test_source_and_output_rects_three_ways(trial):
    """HAS_SOURCE_RECT / HAS_OUTPUT_RECT (the buffer descriptions' rects, mod.rs:279-295): the per-plane kernel (host-interpreted), the oracle and — on the
    luma plane, the field of view inside the frame — the reference's own kernel (host build) must all write the same bytes; pixels outside the output rect
    keep the caller's content."""
    from _refcl import run_reference_cl_host
    rng = np.random.default_rng(700 + trial)
    fmt = ["YUV422P16LE", "NV12", "RGBA", "RGBAF32", "P010LE", "YUV420P"][trial]
    interp = [2, 4, 8][trial % 3]
    w, h = int(rng.integers(60, 160)) * 2, int(rng.integers(40, 100)) * 2
    fr = S.SyntheticFrame(fmt, w, h, seed=900 + trial, fov=0.85, interpolation=interp, horizontal_rs=bool(trial == 4))
    for pl in fr.planes:
        p = pl["params"]
        (pw, ph), (ow, oh) = pl["size"][:2], pl["out_size"][:2]
        x0, y0 = int(rng.integers(1, pw // 5)), int(rng.integers(1, ph // 5))
        p.source_rect[0], p.source_rect[1], p.source_rect[2], p.source_rect[3] = x0, y0, pw - x0 - int(rng.integers(1, pw // 5)), ph - y0 - int(rng.integers(1, ph // 5))
        x0, y0 = int(rng.integers(1, ow // 5)), int(rng.integers(1, oh // 5))
        p.output_rect[0], p.output_rect[1], p.output_rect[2], p.output_rect[3] = x0, y0, ow - x0 - int(rng.integers(1, ow // 5)), oh - y0 - int(rng.integers(1, oh // 5))
        p.flags |= abi.FLAG_HAS_SOURCE_RECT | abi.FLAG_HAS_OUTPUT_RECT
    ref = O.run_frame(fr)
    for i, (a, b) in enumerate(zip(ref, _emu.run_frame_per_plane(fr))):
        assert np.array_equal(a, b), "plane %d: %d bytes differ" % (i, int(np.count_nonzero(a != b)))
    assert np.count_nonzero(ref[0] == 0x5A) > 0                                          # the border outside the output rect kept the 0x5A fill
    name = {"Luma16": "luma16", "Luma8": "luma8", "RGBA8": "rgba8", "RGBAf": "rgbaf"}[fr.planes[0]["pixel_type"]] + "_" + {2: "bilinear", 4: "bicubic", 8: "lanczos4"}[interp] + "_fisheye"
    assert np.array_equal(run_reference_cl_host(name, fr.planes[0], fr.matrices), ref[0])


@pytest.mark.parametrize("fmt,interp", [("YUV422P16LE", 2), ("NV12", 2), ("YUV420P", 4), ("P010LE", 8), ("RGBA", 2), ("RGBAF32", 2), ("YUV444P16LE", 2)])
def test_colour_range_fix_through_the_fused_kernel(fmt, interp):
    """FIX_COLOR_RANGE (cpu_undistort.rs:254-260, :619-621; the render loop raises it for macOS VideoToolbox surfaces, rendering/mod.rs:507-509): the finished pixel —
    sample or background — times the plane's scale (luma: plane_index 0), plus 16 on lanes 0 and 1, before the cast.  Fused since round 5: both forms of the
    kernel's source against the oracle, a zoomed-out frame so that background pixels and edge samples take part."""
    same_as_oracle(S.SyntheticFrame(fmt, 322, 186, seed=84, fov=1.6, interpolation=interp, flags=abi.FLAG_FIX_COLOR_RANGE, limited_range=True,
                                    background_rgba=(0.3, 0.5, 0.7, 1.0)))


@pytest.mark.parametrize("interp", [2, 4, 8])
@pytest.mark.parametrize("fov", [1.0, 1.7])
def test_packed_half_float_pixels_through_the_fused_kernel(interp, fov):
    """RGBAf16 (pixel_formats.rs:227-246: half::f16 to_f32 on load, from_f32 — round to nearest even — on store) on the fused kernel since round 5: the f32 packed path
    with the conversions at the fetch and the store; both kernel forms against the oracle, interior and zoomed-out (background, edge taps)."""
    same_as_oracle(S.SyntheticFrame("RGBAF16", 322, 186, seed=61, fov=fov, interpolation=interp, background_rgba=(0.2, 0.4, 0.6, 0.8)))


# ---- the frame's checksum taken where the pixels leave (GFW_BK_checksum builds: gfw_set_frame_checksums) ----------------------------------

def written_checksum(fr, outs):
    """What gfw_checksum64 of a zero-initialised destination would hold after the frame: every byte the kernel writes (the output pixels: stride padding and the
    planes' untouched tails stay out) times 256^(its address mod 8), summed modulo 2^64 — computed from the planes the interpreter wrote, at THEIR addresses."""
    total = 0
    for pl, arr in zip(fr.planes, outs):
        ow, oh, stride = pl["out_size"]
        row_bytes = ow * pl["params"].bytes_per_pixel
        rows = np.frombuffer(arr, np.uint8)[:oh * stride].reshape(oh, stride)[:, :row_bytes]
        pos = (arr.ctypes.data + np.arange(oh, dtype=np.int64)[:, None] * stride + np.arange(row_bytes, dtype=np.int64)[None, :]) & 7
        for k in range(8):
            total += int(rows[pos == k].astype(np.uint64).sum()) << (8 * k)
    return total & 0xFFFFFFFFFFFFFFFF


@pytest.mark.parametrize("fmt,kw,n,grid", [
    ("YUV422P16LE", dict(), 3, 8),                                   # the branch-free lane-row: pair stores of luma, single chroma values; three frames = two frame changes per wave
    ("YUV422P16LE", dict(interpolation=4), 1, 8),
    ("NV12", dict(), 2, 16),                                         # interleaved 8-bit chroma pairs (a pair may sit anywhere in a word)
    ("YUV420P", dict(interpolation=8), 1, 8),
    ("P010LE", dict(fov=1.6), 2, 8),                                 # out-of-frame pixels: the background goes through the same stores
    ("RGBAF32", dict(), 1, 8),
    ("GBRAPF32LE", dict(), 2, 8),
    ("RGBAF16", dict(), 1, 8),
    ("YUV444P16LE", dict(base_overrides={"lens_correction_amount": 0.5}), 1, 8),      # a generic-model body
])
def test_checksum_taken_at_the_stores_is_the_checksum_of_what_was_written(fmt, kw, n, grid):
    frames = [S.SyntheticFrame(fmt, 200, 120, seed=0xC5 + j, timestamp_ms=1000.0 + 33.3 * j, **kw) for j in range(n)]
    outs, sums = _emu.run_frames(frames, grid=grid, checksums=True)
    for j, (fr, got, s) in enumerate(zip(frames, outs, sums)):
        for p, (a, b) in enumerate(zip(O.run_frame(fr), got)):
            assert np.array_equal(a, b), "frame %d plane %d" % (j, p)
        assert s == written_checksum(fr, got), "frame %d: %016x" % (j, s)


def test_checksum_words_of_frames_a_wave_has_no_tile_of_are_written_too():
    """64 workgroups for a frame of 4 tiles: most waves never see a tile of most frames, and still own a word of every frame in the launch's table of partial
    sums — the interpreter's table starts poisoned, so a word left unwritten shows."""
    frames = [S.SyntheticFrame("YUV422P16LE", 192, 48, seed=0x5C + j, timestamp_ms=1000.0 + 33.3 * j) for j in range(5)]
    outs, sums = _emu.run_frames(frames, grid=64, checksums=True)
    for fr, got, s in zip(frames, outs, sums):
        assert s == written_checksum(fr, got)
