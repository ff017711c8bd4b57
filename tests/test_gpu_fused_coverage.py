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
