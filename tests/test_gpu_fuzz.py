"""Seeded random sweep over the operator's parameter space: format x interpolation x background mode x flags x lens
model x digital lens x readout direction x sizes (odd ones included) x lens-correction amount x stretch x margins.
Each configuration is warped by libgfwarp (fused when eligible, generic otherwise) and by the oracle; results must be
bit-identical.  The seeds are fixed so a failure names its configuration."""
import numpy as np
import pytest

from gyroflow_amd import abi, synthetic as S, warp
import _oracle as O
from test_gpu_parity import assert_plane_equal
from test_gpu_lens_models import PHYSICAL, DIGITAL

pytestmark = pytest.mark.gpu

FORMATS = ["NV12", "NV21", "P010LE", "P210LE", "P416LE", "YUV420P", "YUV420P10LE", "YUV422P12LE", "YUV444P16LE", "YUVA444P10LE",
           "GBRAPF32LE", "GBRPF32LE", "AYUV64LE", "RGB24", "RGBA", "BGRA", "RGB48BE", "RGBA64BE", "RGBAF32", "RGBAF16"]
INTERP = [2, 2, 2, 4, 8, 10, 11, 12, 13]


def random_case(seed):
    rng = np.random.default_rng(1000 + seed)
    fmt = FORMATS[seed % len(FORMATS)]
    w = int(rng.integers(40, 140)) * (1 if rng.random() < 0.3 else 2)
    h = int(rng.integers(30, 100)) * (1 if rng.random() < 0.3 else 2)
    lens = S.gopro_style_lens(w, h)
    model = sorted(PHYSICAL)[int(rng.integers(0, len(PHYSICAL)))] if rng.random() < 0.5 else "opencv_fisheye"
    lens["model"] = model
    lens["k"] = PHYSICAL[model] + [0.0] * (12 - len(PHYSICAL[model]))
    if rng.random() < 0.2:
        lens["r_limit"] = float(rng.uniform(0.8, 2.5))
    ov = {}
    if rng.random() < 0.2:
        d = sorted(DIGITAL)[int(rng.integers(0, len(DIGITAL)))]
        lens["digital"] = d
        ov["digital_lens_params"] = DIGITAL[d]
    if rng.random() < 0.3:
        ov["lens_correction_amount"] = float(rng.uniform(0.0, 1.0))
    if rng.random() < 0.2:
        ov["input_horizontal_stretch"] = float(rng.uniform(0.8, 1.3))
        ov["input_vertical_stretch"] = float(rng.uniform(0.8, 1.3))
    bgm = int(rng.integers(0, 4)) if rng.random() < 0.5 else 0
    ov["background_mode"] = bgm
    if bgm == 3:
        ov["background_margin"] = float(rng.uniform(0.0, 0.3))
        ov["background_margin_feather"] = float(rng.uniform(0.0, 0.3))
    if rng.random() < 0.3:
        ov["translation2d"] = (float(rng.uniform(-20, 20)), float(rng.uniform(-20, 20)))
    flags = 0
    if rng.random() < 0.15:
        flags |= abi.FLAG_FIX_COLOR_RANGE
    kw = dict(seed=int(rng.integers(1, 1 << 20)), lens=lens, fov=float(rng.uniform(0.7, 2.6)),
              readout_ms=0.0 if rng.random() < 0.2 else float(rng.uniform(-25.0, 25.0)),
              interpolation=INTERP[int(rng.integers(0, len(INTERP)))], horizontal_rs=bool(rng.random() < 0.25),
              background_rgba=tuple(float(x) for x in rng.uniform(0.0, 1.0, 4)), base_overrides=ov, flags=flags,
              limited_range=bool(rng.random() < 0.5), stride_align=int(rng.choice([1, 4, 64, 256])))
    if kw["interpolation"] > 8:
        # EWA footprints grow with the local magnification; far outside the image circle of an exploding lens polynomial
        # the reference's unbounded tap loop runs for minutes per frame — keep the EWA cases where it stays tractable
        lens["model"] = "opencv_fisheye"
        lens["k"] = PHYSICAL["opencv_fisheye"] + [0.0] * 8
        lens.pop("digital", None)
        lens["r_limit"] = 0.0                   # rejected neighbours make the finite-difference jacobian explode too
        ov.pop("digital_lens_params", None)
        ov.pop("lens_correction_amount", None)
        kw["fov"] = min(kw["fov"], 1.4)
    if rng.random() < 0.25:
        kw["out_size"] = (max(16, w + 2 * int(rng.integers(-10, 10))), max(16, h + 2 * int(rng.integers(-8, 8))))
    return fmt, w, h, kw


@pytest.mark.parametrize("seed", range(80))
def test_random_configuration_bit_exact(seed):
    fmt, w, h, kw = random_case(seed)
    fr = S.SyntheticFrame(fmt, w, h, **kw)
    ref = O.run_frame(fr)
    got = warp.run_frame(fr)
    backend = warp.last_backend()
    for i, (a, b) in enumerate(zip(ref, got)):
        assert_plane_equal(a, b, fr.planes[i]["pixel_type"], "seed %d %s %dx%d %s plane %d (%s)" % (seed, fmt, w, h, kw, i, backend))
