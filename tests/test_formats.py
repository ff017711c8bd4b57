"""Plane table and background-colour mapping (gyroflow_amd/formats.py) vs the reference's render loop
(rendering/mod.rs:565-649) and `PixelType::from_rgb_color` (pixel_formats.rs:23-49, per-type impls)."""
import numpy as np

from gyroflow_amd import abi, formats as F, synthetic as S


def test_plane_table_matches_render_loop():
    t = F.PLANE_TABLE
    assert [p.pixel_type for p in t["NV12"]] == ["Luma8", "UV8"] and t["NV12"][1].yuv == [1, 2] and t["NV21"][1].yuv == [2, 1]
    for name in ("P010LE", "P016LE", "P210LE", "P216LE", "P410LE", "P416LE"):
        assert [p.pixel_type for p in t[name]] == ["Luma16", "UV16"] and all(p.max_val == 65535.0 for p in t[name])   # :581-585
    assert t["P010LE"][1].sub == (2, 2) and t["P210LE"][1].sub == (2, 1) and t["P410LE"][1].sub == (1, 1)
    for bits, mx in ((10, 1023.0), (12, 4095.0), (14, 16383.0), (16, 65535.0)):
        for ss, sub in (("420", (2, 2)), ("422", (2, 1)), ("444", (1, 1))):
            pl = t["YUV%sP%dLE" % (ss, bits)]
            assert [p.pixel_type for p in pl] == ["Luma16"] * 3 and [p.yuv for p in pl] == [[0], [1], [2]]
            assert pl[1].sub == sub and pl[2].sub == sub and all(p.max_val == mx for p in pl)
    assert len(t["YUVA444P12LE"]) == 4 and t["YUVA444P12LE"][3].yuv == [3] and t["YUVA444P12LE"][0].max_val == 4095.0
    assert "YUVA444P14LE" not in t                                           # :612 lists 10/12/16 only
    assert [p.yuv for p in t["GBRAPF32LE"]] == [[2], [0], [1], [3]] and all(p.max_val == 255.0 and p.pixel_type == "R32f" for p in t["GBRAPF32LE"])
    assert len(t["GBRPF32LE"]) == 3
    assert t["AYUV64LE"][0].yuv == [3, 0, 1, 2] and t["AYUV64LE"][0].pixel_type == "AYUV16"
    assert t["RGB24"][0].pixel_type == "RGB8" and t["RGBA"][0].pixel_type == "RGBA8"
    assert t["RGB48BE"][0].pixel_type == "RGB16" and t["RGBA64BE"][0].pixel_type == "RGBA16"
    assert F.planes_for("bayer_rggb16le") is t["YUV444P16LE"]               # :641-649
    for planes in t.values():
        for p in planes:
            assert p.pixel_type in abi.PIXEL_TYPES


def test_plane_size_rounds_up_like_ffmpeg():
    assert F.plane_size(1920, 1080, (2, 2)) == (960, 540)
    assert F.plane_size(1921, 1081, (2, 2)) == (961, 541)
    assert F.plane_size(1921, 1081, (2, 1)) == (961, 1081)


def test_rgb_to_yuv_known_answers():
    assert F.rgb_to_yuv((0, 0, 0, 0), False) == [0.0, 0.5, 0.5, 0.0]
    y, u, v, a = F.rgb_to_yuv((1, 1, 1, 1), False)
    assert abs(y - 1.0) < 1e-6 and abs(u - 0.5) < 1e-6 and abs(v - 0.5) < 1e-6 and a == 1.0
    y, u, v, _ = F.rgb_to_yuv((0, 0, 0, 0), True)
    assert y == np.float32(16.0) / np.float32(255.0) and abs(u - 128.0 / 255.0) < 1e-6 and u == v
    y, u, v, _ = F.rgb_to_yuv((1, 0, 0, 1), False)                         # pure red: V saturates at 1, U below 0.5
    assert abs(y - 0.2126) < 1e-6 and v == 1.0 and u < 0.5
    assert all(0.0 <= c <= 1.0 for c in F.rgb_to_yuv((7.0, -3.0, 2.0, 9.0), False))
    assert all(isinstance(c, np.float32) for c in F.rgb_to_yuv((0.3, 0.5, 0.7, 1.0), True))


def test_from_rgb_color_per_type():
    c = (0.3, 0.5, 0.7, 0.9)
    yuv = F.rgb_to_yuv(c, False)
    assert F.from_rgb_color("Luma16", c, [1]) == [yuv[1], 0, 0, 0]
    assert F.from_rgb_color("UV8", c, [2, 1]) == [yuv[2], yuv[1], 0, 0]
    assert F.from_rgb_color("AYUV16", c, [3, 0, 1, 2]) == [yuv[3], yuv[0], yuv[1], yuv[2]]
    assert F.from_rgb_color("R32f", c, [2]) == [np.float32(0.7), 0, 0, 0]
    assert F.from_rgb_color("BGRA8", c, []) == [np.float32(0.7), np.float32(0.5), np.float32(0.3), np.float32(0.9)]
    assert F.from_rgb_color("RGBA16", c, []) == [np.float32(x) for x in c]


def test_synthetic_frames_take_their_planes_and_background_from_the_table():
    fr = S.SyntheticFrame("NV21", 64, 48, background_rgba=(0.3, 0.5, 0.7, 1.0), limited_range=True)
    yuv = F.rgb_to_yuv((0.3, 0.5, 0.7, 1.0), True)
    assert fr.planes[0]["params"].background[0] == yuv[0]
    assert (fr.planes[1]["params"].background[0], fr.planes[1]["params"].background[1]) == (yuv[2], yuv[1])
    fr = S.SyntheticFrame("YUV420P10LE", 66, 50)
    assert fr.planes[1]["size"][:2] == (33, 25) and fr.planes[1]["params"].max_pixel_value == 1023.0
