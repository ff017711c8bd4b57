"""Frame formats -> per-plane warp calls (Appendix C).

Host-side mirror of the render loop's plane table `rendering/mod.rs:565-649` (which FFmpeg pixel format becomes which
`PixelType` calls, with which plane size, background-colour component and value range) and of
`PixelType::from_rgb_color` / `rgb_to_yuv` (`stabilization/pixel_formats.rs:23-49` and the per-type impls), so a caller
holding decoded frames — in host memory or, with `GFW_BUF_HIP_DEVICE`, in HBM — can describe them to
`gfw_undistort_frame` without touching pixel data.
"""
from collections import namedtuple

import numpy as np

# pixel_type: PixelType name; sub: (w, h) chroma divisors of the plane relative to the frame; yuv: indices into the
# Y,U,V,A colour for the background (`$yuvi`); max_val: `$max_val` (pixel_value_limit = max_pixel_value)
PlaneSpec = namedtuple("PlaneSpec", "pixel_type sub yuv max_val")


def _yuv_planar(ptype, sub, max_val, alpha=False):
    planes = [PlaneSpec(ptype, (1, 1), [0], max_val), PlaneSpec(ptype, sub, [1], max_val), PlaneSpec(ptype, sub, [2], max_val)]
    if alpha:
        planes.append(PlaneSpec(ptype, (1, 1), [3], max_val))
    return planes


PLANE_TABLE = {
    # rendering/mod.rs:565-576
    "NV12": [PlaneSpec("Luma8", (1, 1), [0], 255.0), PlaneSpec("UV8", (2, 2), [1, 2], 255.0)],
    "NV21": [PlaneSpec("Luma8", (1, 1), [0], 255.0), PlaneSpec("UV8", (2, 2), [2, 1], 255.0)],
    # :577-590 — 10-bit P0xx keeps MSB-aligned 16-bit values (comment at :581-583)
    "P010LE": [PlaneSpec("Luma16", (1, 1), [0], 65535.0), PlaneSpec("UV16", (2, 2), [1, 2], 65535.0)],
    "P016LE": [PlaneSpec("Luma16", (1, 1), [0], 65535.0), PlaneSpec("UV16", (2, 2), [1, 2], 65535.0)],
    "P210LE": [PlaneSpec("Luma16", (1, 1), [0], 65535.0), PlaneSpec("UV16", (2, 1), [1, 2], 65535.0)],
    "P216LE": [PlaneSpec("Luma16", (1, 1), [0], 65535.0), PlaneSpec("UV16", (2, 1), [1, 2], 65535.0)],
    "P410LE": [PlaneSpec("Luma16", (1, 1), [0], 65535.0), PlaneSpec("UV16", (1, 1), [1, 2], 65535.0)],
    "P416LE": [PlaneSpec("Luma16", (1, 1), [0], 65535.0), PlaneSpec("UV16", (1, 1), [1, 2], 65535.0)],
    # :591-597
    "YUV420P": _yuv_planar("Luma8", (2, 2), 255.0),
    "YUVJ420P": _yuv_planar("Luma8", (2, 2), 255.0),
    # :624-635 — all four (three) EXR planes are R32f with max_val 255.0; colour index G,B,R,A -> [2],[0],[1],[3]
    "GBRAPF32LE": [PlaneSpec("R32f", (1, 1), [2], 255.0), PlaneSpec("R32f", (1, 1), [0], 255.0),
                   PlaneSpec("R32f", (1, 1), [1], 255.0), PlaneSpec("R32f", (1, 1), [3], 255.0)],
    "GBRPF32LE": [PlaneSpec("R32f", (1, 1), [2], 255.0), PlaneSpec("R32f", (1, 1), [0], 255.0), PlaneSpec("R32f", (1, 1), [1], 255.0)],
    # :636-640
    "AYUV64LE": [PlaneSpec("AYUV16", (1, 1), [3, 0, 1, 2], 65535.0)],
    "RGB24": [PlaneSpec("RGB8", (1, 1), [], 255.0)],
    "RGBA": [PlaneSpec("RGBA8", (1, 1), [], 255.0)],
    "RGB48BE": [PlaneSpec("RGB16", (1, 1), [], 65535.0)],
    "RGBA64BE": [PlaneSpec("RGBA16", (1, 1), [], 65535.0)],
    # reachable through the library API only (external plugins), not through the render table
    "RGBAF32": [PlaneSpec("RGBAf", (1, 1), [], None)],
    "RGBAF16": [PlaneSpec("RGBAf16", (1, 1), [], None)],
    "BGRA": [PlaneSpec("BGRA8", (1, 1), [], 255.0)],
}
# :598-611 planar high-bit-depth YUV; :612-623 with alpha
for _bits, _max in ((10, 1023.0), (12, 4095.0), (14, 16383.0), (16, 65535.0)):
    for _name, _sub in (("420", (2, 2)), ("422", (2, 1)), ("444", (1, 1))):
        PLANE_TABLE["YUV%sP%dLE" % (_name, _bits)] = _yuv_planar("Luma16", _sub, _max)
for _bits, _max in ((10, 1023.0), (12, 4095.0), (16, 65535.0)):
    PLANE_TABLE["YUVA444P%dLE" % _bits] = _yuv_planar("Luma16", (1, 1), _max, alpha=True)

FALLBACK_FORMAT = "YUV444P16LE"      # :641-649 "All other convert to YUV444P16LE"


def planes_for(fmt):
    """Plane list of an FFmpeg pixel format; unknown formats take the reference's conversion route (:641-649)."""
    return PLANE_TABLE.get(fmt.upper(), PLANE_TABLE[FALLBACK_FORMAT])


def plane_size(width, height, sub):
    """FFmpeg plane dimensions (AV_CEIL_RSHIFT of the frame size; zero_copy::get_plane_size)."""
    return (width + sub[0] - 1) // sub[0], (height + sub[1] - 1) // sub[1]


# ---- background colour: pixel_formats.rs:23-49 in f32 ---------------------------------------------------------------
_f = np.float32
KR, KB = _f(0.2126), _f(0.0722)          # Rec.709
KG = _f(1.0) - KR - KB
US = _f(1.0) / (_f(2.0) - _f(2.0) * KB)
VS = _f(1.0) / (_f(2.0) - _f(2.0) * KR)


def _clamp01(x):
    return min(max(x, _f(0.0)), _f(1.0))          # f32::max(0.0).min(1.0)


def rgb_to_yuv(rgba, is_limited):
    """`rgb_to_yuv` (pixel_formats.rs:23-49), evaluated in float32 in the reference's operation order."""
    v = [_f(c) for c in rgba]
    ret = [
        _clamp01(KR * v[0] + KG * v[1] + KB * v[2]),
        _clamp01((-KR * US) * v[0] + (-KG * US) * v[1] + ((_f(1.0) - KB) * US) * v[2] + _f(0.5)),
        _clamp01(((_f(1.0) - KR) * VS) * v[0] + (-KG * VS) * v[1] + (-KB * VS) * v[2] + _f(0.5)),
        _clamp01(v[3]),
    ]
    if is_limited:
        a = _f(16.0) / _f(255.0)
        ret[0] = a + ret[0] * ((_f(235.0) - _f(16.0)) / _f(255.0))
        ret[1] = a + ret[1] * ((_f(240.0) - _f(16.0)) / _f(255.0))
        ret[2] = a + ret[2] * ((_f(240.0) - _f(16.0)) / _f(255.0))
    return ret


def from_rgb_color(pixel_type, rgba, ind, is_limited=False):
    """`PixelType::from_rgb_color` per type (pixel_formats.rs:77-298): the per-plane KernelParams.background."""
    v = [_f(c) for c in rgba]
    z = _f(0.0)
    if pixel_type in ("Luma8", "Luma16"):
        return [rgb_to_yuv(v, is_limited)[ind[0]], z, z, z]
    if pixel_type in ("UV8", "UV16"):
        yuv = rgb_to_yuv(v, is_limited)
        return [yuv[ind[0]], yuv[ind[1]], z, z]
    if pixel_type == "AYUV16":
        yuv = rgb_to_yuv(v, is_limited)
        return [yuv[ind[0]], yuv[ind[1]], yuv[ind[2]], yuv[ind[3]]]
    if pixel_type == "R32f":
        return [v[ind[0]], z, z, z]
    if pixel_type == "BGRA8":
        return [v[2], v[1], v[0], v[3]]
    return v                                       # RGB8/RGBA8/RGB16/RGBA16/RGBAf/RGBAf16
