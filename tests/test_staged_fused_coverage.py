"""STAGING AREA — fused-kernel paths that are written but have not been through the GPU parity suite yet.

Run on the GPU box with `python -m pytest tests -m gpu_staged`; `-m gpu` (what the round-end driver runs) does not select these,
and without a GPU they are skipped.  A test moves into the regular `-m gpu` files — and its path loses the
GFW_OPT_KERNEL_VARIANT = 7 gate in gfw_api.hip — once it has passed there.

Staged: background mode 3 "margin with feather" (cpu_undistort.rs:576-613) through the fused kernel's generic-model instantiation
(two samples per plane + alpha blend; today the per-plane kernel serves it, bit-exact).
"""
import pytest

from gyroflow_amd import synthetic as S, warp
import _oracle as O
from test_gpu_parity import assert_plane_equal

pytestmark = pytest.mark.gpu_staged

STAGING_VARIANT = 7


def check_staged(fr):
    ref = O.run_frame(fr)
    got = warp.run_frame(fr, variant=STAGING_VARIANT)
    assert warp.last_backend() == "yuv_fused", warp.last_backend()
    for i, (a, b) in enumerate(zip(ref, got)):
        assert_plane_equal(a, b, fr.planes[i]["pixel_type"], "fused background mode 3, plane %d" % i)
    base = warp.run_frame(fr)                                     # the default route is untouched
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
