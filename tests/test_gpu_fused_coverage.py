"""Fused-kernel coverage of background mode 3 "margin with feather" (cpu_undistort.rs:576-613) and of the Sony lens-distortion mesh /
focal-plane-distortion terms (:169-214): both run through the fused kernel's GFW_MODEL_GENERIC_EXTRA instantiation (two samples per
plane + alpha blend; f64 mesh spline) and through the per-plane kernel, and both must equal the oracle bit for bit.  (Validated on
MI355X in round 3 as a staging build — 31 / 31 — and promoted: gpurun_out/r03b/staged.log.)"""
import pytest

from gyroflow_amd import abi, synthetic as S, warp
import _oracle as O
from test_gpu_parity import assert_plane_equal

pytestmark = pytest.mark.gpu



def check_staged(fr):
    ref = O.run_frame(fr)
    got = warp.run_frame(fr)
    assert warp.last_backend() == "yuv_fused"
    for i, (a, b) in enumerate(zip(ref, got)):
        assert_plane_equal(a, b, fr.planes[i]["pixel_type"], "fused background mode 3, plane %d" % i)
    base = warp.run_frame(fr, fused=False)                        # the per-plane kernel
    assert warp.last_backend() == "plane_generic"
    for i, (a, b) in enumerate(zip(ref, base)):
        assert_plane_equal(a, b, fr.planes[i]["pixel_type"], "per-plane background mode 3, plane %d" % i)


@pytest.mark.parametrize("fmt", ["YUV422P16LE", "NV12", "P010", "YUV420P", "YUV444P16LE", "RGBA", "RGBA64", "RGBAF32", "GBRAPF32LE"])
def test_margin_with_feather_fused(fmt):
    ov = {"background_mode": 3, "background_margin": 0.1, "background_margin_feather": 0.05}
    fr = S.SyntheticFrame(fmt, 256, 160, seed=9, fov=2.5, base_overrides=ov, background_rgba=(0.3, 0.5, 0.7, 1.0))
    check_staged(fr)


@pytest.mark.parametrize("interp", [2, 4, 8])
@pytest.mark.parametrize("margin,feather", [(0.0, 0.0), (0.25, 0.2), (0.05, 0.5)])
def test_margin_with_feather_fused_samplers_and_extremes(interp, margin, feather):
    ov = {"background_mode": 3, "background_margin": margin, "background_margin_feather": feather}
    fr = S.SyntheticFrame("YUV422P16LE", 320, 180, seed=21 + interp, fov=1.6, base_overrides=ov, interpolation=interp,
                          background_rgba=(0.1, 0.9, 0.4, 1.0))
    check_staged(fr)


def test_margin_with_feather_fused_with_rolling_shutter_inside_the_frame():
    # fov < 1: every pixel projects inside the source, alpha = 1 except within the feather band of the border
    ov = {"background_mode": 3, "background_margin": 0.1, "background_margin_feather": 0.15}
    fr = S.SyntheticFrame("YUV422P16LE", 640, 360, seed=5, fov=0.9, base_overrides=ov)
    check_staged(fr)


# ---- Sony lens-distortion mesh + focal-plane distortion (cpu_undistort.rs:169-214) through the fused kernel ----------------
def run_frame_with_mesh(fr, mesh, variant):
    import numpy as np
    from gyroflow_amd import abi
    outs = [pl["dst"].copy() for pl in fr.planes]
    bufs = [warp.host_buffers(pl["src"], pl["size"], o, pl["out_size"]) for pl, o in zip(fr.planes, outs)]
    params = [pl["params"] for pl in fr.planes]
    types = [pl["pixel_type"] for pl in fr.planes]
    be = warp.Backend(params[0], types[0], fr.model, fr.digital, bufs[0])
    try:
        if variant:
            be.set_option(abi.OPT_KERNEL_VARIANT, variant)
        be.undistort_frame(bufs, params, types, fr.matrices, mesh=np.asarray(mesh, dtype=np.float32))
    finally:
        be.close()
    return outs


@pytest.mark.parametrize("fmt", ["NV12", "YUV422P16LE", "RGBA64"])
@pytest.mark.parametrize("with_mesh,with_fpd,inverted", [(True, False, False), (True, True, False), (True, True, True), (False, True, False)])
def test_sony_mesh_and_focal_plane_distortion_fused(fmt, with_mesh, with_fpd, inverted):
    from gyroflow_amd import abi
    from test_gpu_lens_models import synthetic_mesh
    w, h = 192, 128
    fr = S.SyntheticFrame(fmt, w, h, seed=47, fov=1.1, flags=abi.FLAG_FRAMEBUFFER_INVERTED if inverted else 0)
    mesh = synthetic_mesh(w, h, with_fpd, with_mesh)
    ref = []
    for pl in fr.planes:
        dst = pl["dst"].copy()
        assert O.undistort_image(pl["src"], pl["size"], dst, pl["out_size"], pl["params"], pl["pixel_type"], fr.model, fr.digital, fr.matrices, mesh=mesh) == 1
        ref.append(dst)
    got = run_frame_with_mesh(fr, mesh, 0)
    assert warp.last_backend() == "yuv_fused"
    for i, (a, b) in enumerate(zip(ref, got)):
        assert_plane_equal(a, b, fr.planes[i]["pixel_type"], "fused mesh, plane %d" % i)
    base = run_frame_with_mesh(fr, mesh, 1)                      # forced per-plane kernel
    assert warp.last_backend() == "plane_generic"
    for i, (a, b) in enumerate(zip(ref, base)):
        assert_plane_equal(a, b, fr.planes[i]["pixel_type"], "per-plane mesh, plane %d" % i)


@pytest.mark.parametrize("fmt", ["YUV422P16LE", "NV12", "RGBA"])
@pytest.mark.parametrize("jit", [0, 2])
@pytest.mark.parametrize("interp", [2, 8])
def test_stretched_clips_take_the_fused_kernel(fmt, jit, interp):
    """input_horizontal / vertical_stretch (anamorphic lens profiles, cpu_undistort.rs:222-223): an IEEE division at the end of the projection — served by the
    fused kernel since round 4 (ahead of time and specialised), no longer by three per-plane launches"""
    ov = {"input_horizontal_stretch": 1.33, "input_vertical_stretch": 0.9}
    fr = S.SyntheticFrame(fmt, 384, 208, seed=77, fov=1.2, base_overrides=ov, interpolation=interp)
    ref = O.run_frame(fr)
    got = warp.run_frame(fr, jit=jit)
    assert warp.last_backend().startswith("yuv_fused") and warp.last_backend().endswith("_jit") == (jit == 2), warp.last_backend()
    for i, (a, b) in enumerate(zip(ref, got)):
        assert_plane_equal(a, b, fr.planes[i]["pixel_type"], "stretched clip, plane %d" % i)


@pytest.mark.parametrize("fmt", ["YUV422P16LE", "NV12", "RGBA"])
@pytest.mark.parametrize("jit", [0, 2])
@pytest.mark.parametrize("rot", [90.0, 180.0, 270.0, 33.5])
def test_rotated_input_takes_the_fused_kernel(fmt, jit, rot):
    """input_rotation (cpu_undistort.rs:484-491: the projected point turned about the frame centre, frame_size turned with it) — fused since round 4"""
    w, h = (256, 384) if rot in (90.0, 270.0) else (384, 256)
    fr = S.SyntheticFrame(fmt, w, h, seed=91, fov=1.3, base_overrides={"input_rotation": rot})
    ref = O.run_frame(fr)
    got = warp.run_frame(fr, jit=jit)
    assert warp.last_backend().startswith("yuv_fused"), warp.last_backend()
    for i, (a, b) in enumerate(zip(ref, got)):
        assert_plane_equal(a, b, fr.planes[i]["pixel_type"], "input rotation %g, plane %d" % (rot, i))


@pytest.mark.parametrize("fmt", ["YUV422P16LE", "NV12", "YUV420P", "RGBA", "GBRAPF32LE"])
@pytest.mark.parametrize("jit", [0, 2])
def test_fill_with_background_takes_the_fused_kernel(fmt, jit):
    """FILL_WITH_BACKGROUND (cpu_undistort.rs:558-561; the render loop raises it for frames outside the trim ranges): one launch writes every plane's background"""
    fr = S.SyntheticFrame(fmt, 322, 186, seed=93, flags=abi.FLAG_FILL_WITH_BACKGROUND, background_rgba=(0.2, 0.4, 0.6, 1.0))
    ref = O.run_frame(fr)
    got = warp.run_frame(fr, jit=jit)
    assert warp.last_backend().startswith("yuv_fused"), warp.last_backend()
    for i, (a, b) in enumerate(zip(ref, got)):
        assert_plane_equal(a, b, fr.planes[i]["pixel_type"], "fill, plane %d" % i)


@pytest.mark.parametrize("fmt", ["YUV422P16LE", "NV12", "YUV420P", "P010LE", "RGBA", "GBRAPF32LE"])
@pytest.mark.parametrize("jit", [0, 2])
@pytest.mark.parametrize("interp", [2, 8])
def test_colour_range_fix_takes_the_fused_kernel(fmt, jit, interp):
    """FIX_COLOR_RANGE (cpu_undistort.rs:254-260, :619-621; rendering/mod.rs:507-509 raises it for macOS VideoToolbox) — fused since round 5: the pixel is scaled and
    offset between the sample and the cast, background pixels included"""
    fr = S.SyntheticFrame(fmt, 322, 186, seed=84, fov=1.6, interpolation=interp, flags=abi.FLAG_FIX_COLOR_RANGE, limited_range=True, background_rgba=(0.3, 0.5, 0.7, 1.0))
    ref = O.run_frame(fr)
    got = warp.run_frame(fr, jit=jit)
    assert warp.last_backend().startswith("yuv_fused"), warp.last_backend()
    for i, (a, b) in enumerate(zip(ref, got)):
        assert_plane_equal(a, b, fr.planes[i]["pixel_type"], "colour-range fix, plane %d" % i)


@pytest.mark.parametrize("jit", [0, 2])
@pytest.mark.parametrize("interp", [2, 4, 8])
@pytest.mark.parametrize("fov", [1.0, 1.7])
def test_packed_half_float_takes_the_fused_kernel(jit, interp, fov):
    """RGBAf16 (pixel_formats.rs:227-246) — fused since round 5: the f32 packed path with v_cvt_f32_f16 / v_cvt_f16_f32 at the fetch and the store"""
    fr = S.SyntheticFrame("RGBAF16", 322, 186, seed=61, fov=fov, interpolation=interp, background_rgba=(0.2, 0.4, 0.6, 0.8))
    ref = O.run_frame(fr)
    got = warp.run_frame(fr, jit=jit)
    assert warp.last_backend().startswith("yuv_fused"), warp.last_backend()
    for i, (a, b) in enumerate(zip(ref, got)):
        assert_plane_equal(a, b, fr.planes[i]["pixel_type"], "RGBAf16, plane %d" % i)


# ---- buffer rects (mod.rs:209-224, 322-323; gpu/mod.rs:17-24) through the fused kernel — round 6 --------------------------------------------------------------
def embed_in_surface(fr, in_org, out_org, pad=(24, 10)):
    """Every plane of `fr` becomes a WINDOW of a larger surface, the way a host hands a plugin a region of its frame buffer: the source pixels sit at `in_org` of a
    buffer `pad` pixels wider / taller than the plane and beyond (source_rect = (x, y, plane_w, plane_h)), the output goes to `out_org` of a larger 0x5A-filled buffer
    (output_rect likewise); chroma planes take the origins divided by their subsampling.  Bytes outside the output rect must keep their 0x5A."""
    import numpy as np
    lw, lh, loh = fr.planes[0]["size"][0], fr.planes[0]["size"][1], fr.planes[0]["out_size"][1]
    for pl in fr.planes:
        pw, ph, stride = pl["size"]
        ow, oh, ostride = pl["out_size"]
        bpp = pl["params"].bytes_per_pixel
        sub = max(1, lw // pw)
        ix, iy = in_org[0] // sub, in_org[1] // max(1, lh // ph)
        ox, oy = out_org[0] // sub, out_org[1] // max(1, loh // oh)
        bw, bh = pw + ix + pad[0], ph + iy + pad[1]
        bstride = S.align(bw * bpp, 64)
        big = np.random.default_rng(pl["seed"]).integers(0, 256, size=bstride * bh, dtype=np.uint8)       # what lies around the window is NOT background: it must never be sampled
        src2d = pl["src"].reshape(ph, stride)[:, :pw * bpp]
        big.reshape(bh, bstride)[iy:iy + ph, ix * bpp:(ix + pw) * bpp] = src2d
        obw, obh = ow + ox + pad[0], oh + oy + pad[1]
        obstride = S.align(obw * bpp, 64)
        p = pl["params"].copy()
        p.stride, p.output_stride = bstride, obstride
        p.source_rect[0], p.source_rect[1], p.source_rect[2], p.source_rect[3] = ix, iy, pw, ph
        p.output_rect[0], p.output_rect[1], p.output_rect[2], p.output_rect[3] = ox, oy, ow, oh
        p.flags |= abi.FLAG_HAS_SOURCE_RECT | abi.FLAG_HAS_OUTPUT_RECT
        pl.update(src=big, size=(bw, bh, bstride), dst=np.full(obstride * obh, 0x5A, dtype=np.uint8), out_size=(obw, obh, obstride), params=p)
    return fr


@pytest.mark.parametrize("fmt", ["YUV422P16LE", "NV12", "P010", "YUV420P", "RGBA", "RGBA64", "RGBAF32", "GBRAPF32LE"])
@pytest.mark.parametrize("interp", [2, 4, 8])
@pytest.mark.parametrize("jit", [0, 2])
def test_windows_of_larger_surfaces_take_the_fused_kernel(fmt, interp, jit):
    """Source and output rects — a plane that is a window of a larger buffer — ran through the per-plane kernel until round 6 (373 us per C2 frame: 2 % of the
    roofline).  The fused kernel now works on the rects as planes (pointers rebased by the host) and reproduces the one thing that does not rebase: the source
    coordinate's `+ source_rect origin` before the 1/32-pixel binning (cpu_undistort.rs:510-515; a coordinate near 2000 + an origin of 36 rounds differently from the
    coordinate alone).  A fov > 1 frame samples across the window's edges: what surrounds the window in the buffer is noise, what the kernel must use is background.
    (Origins that leave every plane's first pixel dword-aligned: the bicubic / Lanczos4 taps of 8- and 16-bit planes are fetched as aligned dwords from the plane's
    start, and a window that breaks that — a chroma origin of 18 bytes — goes the per-plane way, as an odd sub-buffer always has.)"""
    import numpy as np
    fr = embed_in_surface(S.SyntheticFrame(fmt, 384, 208, seed=0x6EC7 + interp, fov=1.35, interpolation=interp, background_rgba=(0.2, 0.6, 0.4, 1.0)), (40, 14), (20, 6))
    ref = O.run_frame(fr)
    got = warp.run_frame(fr, jit=jit)
    assert warp.last_backend().startswith("yuv_fused") and warp.last_backend().endswith("_jit") == (jit == 2), warp.last_backend()
    for i, (a, b) in enumerate(zip(ref, got)):
        assert_plane_equal(a, b, fr.planes[i]["pixel_type"], "window of a surface, plane %d" % i)
        assert np.count_nonzero(a == 0x5A) > 0
    base = warp.run_frame(fr, fused=False)
    assert warp.last_backend() == "plane_generic"
    for i, (a, b) in enumerate(zip(ref, base)):
        assert_plane_equal(a, b, fr.planes[i]["pixel_type"], "per-plane window, plane %d" % i)


@pytest.mark.parametrize("origin", [(0, 0), (2, 0), (0, 2), (1022, 600)])
def test_window_origins_including_large_ones(origin):
    """The origin enters a float addition: large origins change which 1/32-pixel bin a coordinate falls into (the whole point of adding it as the reference does);
    zero components take the path without the addition."""
    fr = embed_in_surface(S.SyntheticFrame("YUV422P16LE", 320, 192, seed=0x0719, fov=0.95), origin, (origin[0] // 2 * 2, origin[1]), pad=(6, 4))
    ref = O.run_frame(fr)
    for jit in (0, 2):
        got = warp.run_frame(fr, jit=jit)
        assert warp.last_backend().startswith("yuv_fused"), warp.last_backend()
        for i, (a, b) in enumerate(zip(ref, got)):
            assert_plane_equal(a, b, fr.planes[i]["pixel_type"], "origin %s, plane %d" % (origin, i))
