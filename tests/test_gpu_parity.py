"""GPU parity: libgfwarp (HIP, through the C ABI) vs the oracle on the same seeded inputs.

Bar: bit-exact for u8/u16 planes; f32 planes within 1 ULP (the tests below actually find 0 ULP,
and assert <= 1).  Sizes are chosen so the oracle finishes in seconds.
"""
import numpy as np
import pytest

from gyroflow_amd import abi, synthetic as S, warp
import _oracle as O

pytestmark = pytest.mark.gpu


def ulp_diff(a, b):
    """Max ULP distance between two float32 arrays (NaNs must coincide)."""
    a = np.ascontiguousarray(a, dtype=np.float32)
    b = np.ascontiguousarray(b, dtype=np.float32)
    assert np.array_equal(np.isnan(a), np.isnan(b))
    ia = a.view(np.int32).astype(np.int64)
    ib = b.view(np.int32).astype(np.int64)
    ia = np.where(ia < 0, -(ia & 0x7FFFFFFF), ia)
    ib = np.where(ib < 0, -(ib & 0x7FFFFFFF), ib)
    d = np.abs(ia - ib)
    d[np.isnan(a)] = 0
    return int(d.max()) if d.size else 0


def assert_plane_equal(ref, got, pixel_type, where=""):
    dt = abi.PIXEL_TYPES[pixel_type][1]
    if np.dtype(dt).kind == "f" and np.dtype(dt).itemsize == 4:
        u = ulp_diff(ref.view(np.float32), got.view(np.float32))
        assert u <= 1, "%s: f32 planes differ by %d ULP" % (where, u)
    else:
        if not np.array_equal(ref, got):
            bad = np.flatnonzero(ref != got)
            raise AssertionError("%s: %d bytes differ (first at %d: ref %d got %d)" % (where, bad.size, bad[0], ref[bad[0]], got[bad[0]]))


def check_frame(fr, expect_backend=None):
    """Oracle vs (a) the frame entry point (fused kernel when eligible) and (b) the generic per-plane kernel."""
    ref = O.run_frame(fr)
    got = warp.run_frame(fr, fused=True)
    backend = warp.last_backend()
    for i, (a, b) in enumerate(zip(ref, got)):
        assert_plane_equal(a, b, fr.planes[i]["pixel_type"], "%s plane %d" % (backend, i))
    if expect_backend is not None:
        assert backend.startswith(expect_backend), backend
    if backend.startswith("yuv_fused_p1"):              # also the fused kernel with the exact first pass
        got3 = warp.run_frame(fr, variant=2)
        assert warp.last_backend() == "yuv_fused"
        for i, (a, b) in enumerate(zip(ref, got3)):
            assert_plane_equal(a, b, fr.planes[i]["pixel_type"], "yuv_fused(exact pass 1) plane %d" % i)
    if backend != "plane_generic":
        got2 = warp.run_frame(fr, fused=False)
        assert warp.last_backend() == "plane_generic"
        for i, (a, b) in enumerate(zip(ref, got2)):
            assert_plane_equal(a, b, fr.planes[i]["pixel_type"], "plane_generic plane %d" % i)
    return ref, got


@pytest.mark.parametrize("fmt", ["NV12", "P010", "P210", "YUV420P", "YUV420P10LE", "YUV422P16LE", "YUV444P16LE",
                                 "GBRAPF32LE", "RGBA", "RGBA64", "RGBAF32"])
def test_formats_rolling_shutter(fmt):
    check_frame(S.SyntheticFrame(fmt, 320, 192, seed=11))


@pytest.mark.parametrize("interp", [2, 4, 8, 10, 11, 12, 13])
@pytest.mark.parametrize("ptype_fmt", ["YUV422P16LE", "NV12", "RGBAF32"])
def test_interpolations(interp, ptype_fmt):
    check_frame(S.SyntheticFrame(ptype_fmt, 192, 128, seed=5, interpolation=interp))


def test_per_plane_calls_equal_frame_call():
    fr = S.SyntheticFrame("YUV422P16LE", 256, 160, seed=3)
    a = warp.run_frame(fr, fused=True)
    assert warp.last_backend().startswith("yuv_fused")
    b = warp.run_frame(fr, fused=False, per_plane=True)
    c = warp.run_frame(fr, fused=True, per_plane=True)
    for x, y, z in zip(a, b, c):
        assert np.array_equal(x, y) and np.array_equal(x, z)


@pytest.mark.parametrize("fmt", ["NV12", "P010", "P210", "YUV420P", "YUV422P16LE", "YUV444P16LE", "YUV420P10LE"])
def test_fused_kernel_is_used_for_yuv(fmt):
    check_frame(S.SyntheticFrame(fmt, 320, 192, seed=13), expect_backend="yuv_fused")


def test_fused_kernel_out_of_frame_and_edges():
    # zoomed-out view: large background regions + taps straddling every source edge
    check_frame(S.SyntheticFrame("YUV422P16LE", 320, 192, seed=17, fov=3.0, background_rgba=(0.25, 0.5, 0.75, 1.0)), expect_backend="yuv_fused")
    check_frame(S.SyntheticFrame("NV12", 320, 192, seed=17, fov=3.0, background_rgba=(0.25, 0.5, 0.75, 1.0)), expect_backend="yuv_fused")


@pytest.mark.parametrize("bgmode", [1, 2])
def test_fused_kernel_edge_repeat_and_mirror(bgmode):
    ov = {"background_mode": bgmode}
    check_frame(S.SyntheticFrame("YUV422P16LE", 320, 192, seed=27, fov=2.2, base_overrides=ov), expect_backend="yuv_fused")
    check_frame(S.SyntheticFrame("NV12", 320, 192, seed=27, fov=2.2, base_overrides=ov, interpolation=8), expect_backend="yuv_fused")


def test_fused_kernel_odd_sizes():
    check_frame(S.SyntheticFrame("YUV422P16LE", 322, 190, seed=19), expect_backend="yuv_fused")
    check_frame(S.SyntheticFrame("YUV420P", 130, 66, seed=19), expect_backend="yuv_fused")


def test_fused_kernel_no_rolling_shutter_and_horizontal():
    check_frame(S.SyntheticFrame("YUV422P16LE", 256, 160, seed=23, readout_ms=0.0), expect_backend="yuv_fused")
    check_frame(S.SyntheticFrame("YUV422P16LE", 256, 160, seed=23, horizontal_rs=True), expect_backend="yuv_fused")


def test_c1_1080p_nv12_constant_quaternion():
    q = S.quat_from_euler_deg(5.0, 2.0, 3.0)
    fr = S.SyntheticFrame("NV12", 1920, 1080, seed=0x9F10, readout_ms=0.0, constant_quat=q)
    assert fr.matrices.shape[0] == 1
    check_frame(fr)


def test_c2_like_quarter_size():
    check_frame(S.SyntheticFrame("YUV422P16LE", 960, 540, seed=0x9F10))


@pytest.mark.parametrize("bgmode", [0, 1, 2, 3])
def test_background_modes_with_zoomed_out_view(bgmode):
    ov = {"background_mode": bgmode, "background_margin": 0.1, "background_margin_feather": 0.05}
    fr = S.SyntheticFrame("YUV422P16LE", 256, 160, seed=9, fov=2.5, base_overrides=ov, background_rgba=(0.3, 0.5, 0.7, 1.0))
    check_frame(fr)


def test_horizontal_rolling_shutter():
    check_frame(S.SyntheticFrame("YUV422P16LE", 256, 160, seed=4, horizontal_rs=True))


def test_adaptive_zoom_crop_f32():
    ov = {"translation2d": (13.25, -7.5)}
    check_frame(S.SyntheticFrame("RGBAF32", 320, 192, seed=21, fov=0.82, base_overrides=ov))
    check_frame(S.SyntheticFrame("GBRAPF32LE", 320, 192, seed=21, fov=0.82, base_overrides=ov))


def test_fill_with_background_flag():
    fr = S.SyntheticFrame("NV12", 128, 64, seed=2, flags=abi.FLAG_FILL_WITH_BACKGROUND, background_rgba=(0.2, 0.4, 0.6, 1.0))
    check_frame(fr)


def test_fix_color_range_flag():
    check_frame(S.SyntheticFrame("NV12", 128, 64, seed=2, flags=abi.FLAG_FIX_COLOR_RANGE))


def test_lens_correction_amount_blend():
    ov = {"lens_correction_amount": 0.4}
    check_frame(S.SyntheticFrame("YUV422P16LE", 256, 160, seed=6, base_overrides=ov))


def test_underwater_refraction():
    ov = {"light_refraction_coefficient": 1.33}
    check_frame(S.SyntheticFrame("YUV422P16LE", 256, 160, seed=6, base_overrides=ov))
    ov = {"light_refraction_coefficient": 1.33, "lens_correction_amount": 0.5}
    check_frame(S.SyntheticFrame("YUV422P16LE", 256, 160, seed=6, base_overrides=ov))
