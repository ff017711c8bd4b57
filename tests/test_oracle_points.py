"""Oracle self-consistency for the inverse point map (`undistort_points`, cpu_undistort.rs:652-858): it must invert the
forward coordinate stage (`rotate_and_distort`, the STMap 'undist' closure) that the same oracle restates.  The reference
ships no vectors for this path, so this round trip is the pin."""
import numpy as np
import pytest

from gyroflow_amd import abi, synthetic as S
import _oracle as O

CASES = {
    "opencv_fisheye": [0.045, 0.02, -0.02, 0.006],
    "opencv_standard": [0.1, -0.05, 0.001, 0.002, 0.01, 0.0, 0.0, 0.0],
    "poly5": [0.08, -0.02],
    "sony": [1.0, 0.01, -0.05, 0.02, 0.0, 0.0],
    "gopro": [0.0, 1.0, 0.01, -0.12, 0.02, 0.01, -0.004],
}


def points_params(fr):
    """The KernelParams `undistort_points` builds (cpu_undistort.rs:669-681): only sizes, f, c, k, digital params, refraction."""
    src = fr.planes[0]["params"]
    kp = abi.KernelParams()
    kp.width, kp.height, kp.output_width, kp.output_height = fr.width, fr.height, fr.out_size[0], fr.out_size[1]
    for i in range(2):
        kp.f[i], kp.c[i] = src.f[i], src.c[i]
    for i in range(12):
        kp.k[i] = src.k[i]
    for i in range(16):
        kp.digital_lens_params[i] = src.digital_lens_params[i]
    kp.light_refraction_coefficient = src.light_refraction_coefficient
    kp.lens_correction_amount = 1.0
    return kp


@pytest.mark.parametrize("model", sorted(CASES))
@pytest.mark.parametrize("rolling", [False, True])
def test_inverse_map_inverts_forward_map(model, rolling):
    w, h = 192, 108
    lens = S.gopro_style_lens(w, h)
    lens["model"] = model
    lens["k"] = CASES[model] + [0.0] * (12 - len(CASES[model]))
    fr = S.SyntheticFrame("YUV422P16LE", w, h, seed=9, lens=lens, fov=1.5, readout_ms=16.0 if rolling else 0.0)
    kp = fr.planes[0]["params"].copy()
    kp.flags = 0
    fwd = O.stmap_undistort(kp, fr.model, 0, fr.matrices, w, h)           # output pixel -> source point
    ys, xs = np.mgrid[8:h - 8:7, 8:w - 8:7]
    src = fwd[ys, xs].reshape(-1, 2)
    ok = (src[:, 0] != 0) | (src[:, 1] != 0)
    rows = np.clip(np.round(src[:, 1]).astype(np.int64), 0, fr.rotations.shape[0] - 1) if rolling else np.zeros(len(src), dtype=np.int64)
    back = O.undistort_points(points_params(fr), fr.model, 0, fr.rotations[rows], points=src, index_mode=1)
    want = np.stack([xs.reshape(-1), ys.reshape(-1)], axis=1).astype(np.float32)
    err = np.abs(back - want)[ok]
    assert ok.sum() > 0.8 * len(src)
    # rolling shutter: the forward map picks its row from the mid-row estimate, the inverse from the point itself
    assert err.max() < (0.05 if rolling else 2e-3), err.max()


def test_inverse_map_failed_lens_inverse_and_empty_input():
    w, h = 64, 48
    lens = S.gopro_style_lens(w, h)
    lens["model"] = "poly3"
    lens["k"] = [0.9] + [0.0] * 11
    fr = S.SyntheticFrame("NV12", w, h, seed=3, lens=lens, readout_ms=0.0)
    kp = points_params(fr)
    pts = np.array([[1e7, 1e7], [w / 2 + 3, h / 2 - 2]], dtype=np.float32)
    out = O.undistort_points(kp, fr.model, 0, fr.rotations, points=pts)
    ok, _, _ = O.undistort_point(fr.model, kp, (pts[0, 0] - kp.c[0]) / kp.f[0], (pts[0, 1] - kp.c[1]) / kp.f[1])
    if not ok:
        assert out[0, 0] == -1000000.0 and out[0, 1] == -1000000.0          # cpu_undistort.rs:855
    assert abs(out[1, 0]) < 1000.0
    assert O.undistort_points(kp, fr.model, 0, fr.rotations, points=np.zeros((0, 2), np.float32)).shape == (0, 2)


@pytest.mark.parametrize("model", ["opencv_fisheye", "opencv_standard", "sony"])
@pytest.mark.parametrize("lca", [0.3, 0.75])
def test_lens_correction_branch_inverts_the_render_blend(model, lca):
    """lens_correction_amount < 1: `undistort_points` (cpu_undistort.rs:785-851) must undo the blend `undistort_coord`
    applies on the way in (:429-460).  Newton stops at |g| < 0.02 px, so the round trip is good to a few hundredths."""
    w, h = 192, 108
    lens = S.gopro_style_lens(w, h)
    lens["model"] = model
    lens["k"] = CASES[model] + [0.0] * (12 - len(CASES[model]))
    fr = S.SyntheticFrame("YUV422P16LE", w, h, seed=9, lens=lens, fov=1.3, readout_ms=0.0, base_overrides={"lens_correction_amount": lca})
    kp = fr.planes[0]["params"]
    pts, want = [], []
    for y in range(10, h - 10, 11):
        for x in range(10, w - 10, 13):
            ok, u, v = O.undistort_coord(kp, fr.model, 0, fr.matrices, float(x), float(y))
            if ok:
                pts.append((u, v)); want.append((x, y))
    assert len(pts) > 50
    pp = points_params(fr)
    pp.lens_correction_amount = lca
    pp.fov = kp.fov
    back = O.undistort_points(pp, fr.model, 0, fr.rotations, points=np.array(pts, np.float32))
    err = np.abs(back - np.array(want, np.float32))
    assert err.max() < 0.06, err.max()
    # and the branch really is taken: with lens_correction_amount = 1 the same points land elsewhere
    pp.lens_correction_amount = 1.0
    plain = O.undistort_points(pp, fr.model, 0, fr.rotations, points=np.array(pts, np.float32))
    assert np.abs(plain - back).max() > 0.1
