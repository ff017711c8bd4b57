"""Known-answer tests that pin the oracle (oracle/gfw_oracle.c) analytically.

The reference ships no tests or golden vectors for this path, so these
closed-form cases are what anchors the restatement:
  * identity warp (matrix = inv(K), zero distortion) reproduces the input on the integer
    grid, because phase-0 tap weights are {1, 0, ...} (cpu_undistort.rs:14,23,37);
  * an integer translation shifts the image and fills the uncovered band with background;
  * FILL_WITH_BACKGROUND writes from_float(background * max_pixel_value) (cpu_undistort.rs:558-561);
  * w <= 0 (behind the camera) yields background everywhere (cpu_undistort.rs:138,554);
  * `as u8/u16` stores truncate and saturate, NaN -> 0 (pixel_formats.rs from_float);
  * stride padding is never written (cpu_undistort.rs:551).
"""
import numpy as np
import pytest

from gyroflow_amd import abi, synthetic as S
import _oracle as O


def identity_setup(ptype, w=64, h=48, interp=2, shift=(0, 0), max_val=None, bg=(0, 0, 0, 0), flags=0, wsign=1.0):
    lens = {"model": "opencv_fisheye", "f": (64.0, 64.0), "c": (w / 2.0, h / 2.0), "k": [0.0] * 12, "r_limit": 0.0}
    base = S.base_kernel_params(lens, 1.0, 1)
    src, stride = S.make_plane_buffer(w, h, ptype, 1234, max_val)
    _, dt, count, _ = abi.PIXEL_TYPES[ptype]
    dst = np.full(stride * h, 0x5A, dtype=np.uint8)
    kp = S.plane_kernel_params(base, ptype, (w, h), (w, h), (w, h, stride, None, None), (w, h, stride, None, None),
                               interpolation=interp, flags=flags, background=bg, max_val=max_val)
    k = np.array([[64.0, 0, w / 2.0], [0, 64.0, h / 2.0], [0, 0, 1.0]])
    inv = np.linalg.inv(k)
    inv[0, 2] += shift[0] / 64.0
    inv[1, 2] += shift[1] / 64.0
    m = np.zeros((1, 14), dtype=np.float32)
    m[0, :9] = (inv * wsign).reshape(9) if wsign != 1.0 else inv.reshape(9)
    if wsign != 1.0:
        m[0, :9] = inv.reshape(9)
        m[0, 8] = -1.0
    return src, dst, stride, kp, m, np.dtype(dt).itemsize * count


def view(buf, h, stride, w, bpp):
    return buf.reshape(h, stride)[:, : w * bpp]


@pytest.mark.parametrize("ptype", list(abi.PIXEL_TYPES))
@pytest.mark.parametrize("interp", [2, 4, 8])
def test_identity_warp_reproduces_input(ptype, interp):
    w, h = 64, 48
    src, dst, stride, kp, m, bpp = identity_setup(ptype, w, h, interp)
    assert O.undistort_image(src, (w, h, stride), dst, (w, h, stride), kp, ptype, 1, 0, m) == 1
    a, b = view(src, h, stride, w, bpp), view(dst, h, stride, w, bpp)
    if interp == 2:
        assert np.array_equal(a, b)
    else:
        # phase-0 weights of the wider kernels are {0,1,0,0}/{0,0,0,1,0,..}: interior pixels (whose taps are
        # all inside the source) reproduce exactly; border pixels mix in background*0 = 0 -> still exact.
        assert np.array_equal(a, b)
    # padding bytes untouched
    assert np.all(dst.reshape(h, stride)[:, w * bpp:] == 0x5A)


def test_integer_translation_and_background():
    w, h = 64, 48
    src, dst, stride, kp, m, bpp = identity_setup("Luma16", w, h, 2, shift=(5, -3), bg=(0.25, 0, 0, 0))
    assert O.undistort_image(src, (w, h, stride), dst, (w, h, stride), kp, "Luma16", 1, 0, m) == 1
    a = view(src, h, stride, w, bpp).view("<u2")
    b = view(dst, h, stride, w, bpp).view("<u2")
    bgv = int(np.float32(0.25) * np.float32(65535.0))          # from_float truncates: 16383
    exp = np.full((h, w), bgv, dtype=np.uint16)
    # out(x,y) = in(x+5, y-3)
    exp[3:, : w - 5] = a[: h - 3, 5:]
    assert np.array_equal(b, exp)


@pytest.mark.parametrize("ptype,bg,exp", [
    ("Luma8", (0.5, 0, 0, 0), [127]), ("UV16", (0.5, 1.5, 0, 0), [32767, 65535]),
    ("RGBA8", (-1.0, 0.2, 1.0, float("nan")), [0, 51, 255, 0]), ("R32f", (0.5, 0, 0, 0), [np.float32(0.5)]),
])
def test_fill_with_background(ptype, bg, exp):
    w, h = 64, 48
    src, dst, stride, kp, m, bpp = identity_setup(ptype, w, h, 2, bg=bg, flags=abi.FLAG_FILL_WITH_BACKGROUND)
    assert O.undistort_image(src, (w, h, stride), dst, (w, h, stride), kp, ptype, 1, 0, m) == 1
    _, dt, count, _ = abi.PIXEL_TYPES[ptype]
    b = view(dst, h, stride, w, bpp).view(dt).reshape(h, w, count)
    assert np.array_equal(b, np.broadcast_to(np.array(exp, dtype=dt), (h, w, count)))


def test_behind_camera_is_background():
    w, h = 64, 48
    src, dst, stride, kp, m, bpp = identity_setup("Luma8", w, h, 2, bg=(0.1, 0, 0, 0), wsign=-1.0)
    assert O.undistort_image(src, (w, h, stride), dst, (w, h, stride), kp, "Luma8", 1, 0, m) == 1
    assert np.all(view(dst, h, stride, w, bpp) == int(np.float32(0.1) * np.float32(255.0)))


def test_half_pixel_shift_is_exact_average():
    # shift by 0.5 px in x: phase 16 weights {0.5,0.5}: out = (a+b)/2 truncated
    w, h = 64, 48
    src, dst, stride, kp, m, bpp = identity_setup("Luma16", w, h, 2)
    m[0, 2] += np.float32(0.5 / 64.0)
    assert O.undistort_image(src, (w, h, stride), dst, (w, h, stride), kp, "Luma16", 1, 0, m) == 1
    a = view(src, h, stride, w, bpp).view("<u2").astype(np.float64)
    b = view(dst, h, stride, w, bpp).view("<u2")
    exp = np.floor((a[:, :-1] + a[:, 1:]) / 2.0).astype(np.uint16)
    assert np.array_equal(b[:, :-1], exp)
    assert np.array_equal(b[:, -1], np.floor(a[:, -1] / 2.0).astype(np.uint16))   # right tap outside -> bg 0


def test_pixel_value_limit_clamps():
    w, h = 64, 48
    src, dst, stride, kp, m, bpp = identity_setup("Luma16", w, h, 2, max_val=1023.0)
    s = view(src, h, stride, w, bpp).view("<u2")
    s[:] = 60000                                                  # above the 10-bit limit
    assert O.undistort_image(src, (w, h, stride), dst, (w, h, stride), kp, "Luma16", 1, 0, m) == 1
    assert np.all(view(dst, h, stride, w, bpp).view("<u2") == 1023)


def test_chroma_plane_maps_through_full_res_coords():
    # 4:2:2 chroma plane run with full-res Stabilization size (rendering/mod.rs:514): identity stays identity.
    w, h = 64, 48
    lens = {"model": "opencv_fisheye", "f": (64.0, 64.0), "c": (w / 2.0, h / 2.0), "k": [0.0] * 12}
    base = S.base_kernel_params(lens, 1.0, 1)
    pw = w // 2
    src, stride = S.make_plane_buffer(pw, h, "Luma16", 77)
    dst = np.full(stride * h, 0x5A, dtype=np.uint8)
    kp = S.plane_kernel_params(base, "Luma16", (w, h), (w, h), (pw, h, stride, None, None), (pw, h, stride, None, None))
    assert kp.flags & abi.FLAG_HAS_SOURCE_RECT and kp.flags & abi.FLAG_HAS_OUTPUT_RECT
    m = np.zeros((1, 14), dtype=np.float32)
    m[0, :9] = np.linalg.inv(np.array([[64.0, 0, w / 2.0], [0, 64.0, h / 2.0], [0, 0, 1.0]])).reshape(9)
    assert O.undistort_image(src, (pw, h, stride), dst, (pw, h, stride), kp, "Luma16", 1, 0, m) == 1
    assert np.array_equal(view(src, h, stride, pw, 2), view(dst, h, stride, pw, 2))
