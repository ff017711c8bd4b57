#!/usr/bin/env python3
"""Generate tests/golden/golden.json: CRC32 of every output plane the ORACLE produces for a fixed list of seeded
synthetic configurations, plus a few raw pixel rows.  The reference ships no golden vectors for this path
(SURVEY.md section 4); these freeze the oracle's behaviour so that (a) any later edit of oracle/gfw_oracle.c that
changes results is caught on CPU and (b) the GPU path is checked against numbers that were committed before it ran.

Run from the repo root:  python tests/golden/make_golden.py
"""
import json
import os
import sys
import zlib

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from gyroflow_amd import abi, synthetic as S  # noqa: E402
import _oracle as O  # noqa: E402

CASES = {
    "c1_nv12_1080p_constquat": dict(fmt="NV12", w=1920, h=1080, seed=0x9F10, readout_ms=0.0, quat=(5.0, 2.0, 3.0)),
    "c2_yuv422p16_480x270_rs": dict(fmt="YUV422P16LE", w=480, h=270, seed=0x9F10),
    "p010_320x192_rs": dict(fmt="P010", w=320, h=192, seed=11),
    "yuv420p10_320x192_rs": dict(fmt="YUV420P10LE", w=320, h=192, seed=11),
    "rgbaf32_crop_320x192": dict(fmt="RGBAF32", w=320, h=192, seed=21, fov=0.82, ov={"translation2d": (13.25, -7.5)}),
    "gbrapf32_crop_320x192": dict(fmt="GBRAPF32LE", w=320, h=192, seed=21, fov=0.82, ov={"translation2d": (13.25, -7.5)}),
    "yuv422p16_lanczos4": dict(fmt="YUV422P16LE", w=192, h=128, seed=5, interp=8),
    "yuv422p16_bicubic": dict(fmt="YUV422P16LE", w=192, h=128, seed=5, interp=4),
    "yuv422p16_ewa_robidoux": dict(fmt="YUV422P16LE", w=192, h=128, seed=5, interp=11),
    "yuv422p16_zoomout_bg": dict(fmt="YUV422P16LE", w=256, h=160, seed=9, fov=2.5, bg=(0.3, 0.5, 0.7, 1.0)),
    "yuv422p16_mirror": dict(fmt="YUV422P16LE", w=256, h=160, seed=9, fov=2.5, ov={"background_mode": 2}),
    "yuv422p16_lca04": dict(fmt="YUV422P16LE", w=256, h=160, seed=6, ov={"lens_correction_amount": 0.4}),
    "nv12_horizontal_rs": dict(fmt="NV12", w=256, h=160, seed=4, hrs=True),
}


# coordinate maps (STMap export): CRC32 of the oracle's "undist" (stmap.rs:87-109) and "dist"
# (stmap.rs:123-127 -> undistort_points) f32 maps
MAP_CASES = {
    "stmap_fisheye_384x216_rs": dict(fmt="YUV422P16LE", w=384, h=216, seed=5, fov=1.4),
    "stmap_fisheye_256x160_hrs": dict(fmt="NV12", w=256, h=160, seed=4, fov=1.2, hrs=True),
}


def points_params(fr):
    """The KernelParams `undistort_points` builds (cpu_undistort.rs:669-681)."""
    src = fr.planes[0]["params"]
    kp = abi.KernelParams()
    kp.width, kp.height, kp.output_width, kp.output_height = fr.width, fr.height, fr.out_size[0], fr.out_size[1]
    for i in range(2):
        kp.f[i], kp.c[i] = src.f[i], src.c[i]
    for i in range(12):
        kp.k[i] = src.k[i]
    kp.light_refraction_coefficient = src.light_refraction_coefficient
    kp.lens_correction_amount = 1.0
    return kp


def map_inputs(case):
    fr = build(case)
    kp = fr.planes[0]["params"].copy()
    kp.flags = abi.FLAG_HORIZONTAL_RS if case.get("hrs") else 0
    mode = abi.POINT_INDEX_PER_COLUMN if case.get("hrs") else abi.POINT_INDEX_PER_ROW
    return fr, kp, points_params(fr), mode


def build(case):
    q = S.quat_from_euler_deg(*case["quat"]) if "quat" in case else None
    return S.SyntheticFrame(case["fmt"], case["w"], case["h"], seed=case["seed"], fov=case.get("fov", 1.0),
                            readout_ms=case.get("readout_ms", 16.0), interpolation=case.get("interp", 2),
                            constant_quat=q, horizontal_rs=case.get("hrs", False),
                            background_rgba=case.get("bg", (0.0, 0.0, 0.0, 0.0)), base_overrides=case.get("ov"))


def main():
    out = {}
    for name, case in CASES.items():
        fr = build(case)
        planes = O.run_frame(fr)
        entry = {"planes": [zlib.crc32(p.tobytes()) for p in planes],
                 "src": [zlib.crc32(pl["src"].tobytes()) for pl in fr.planes],
                 "matrices": zlib.crc32(fr.matrices.tobytes())}
        pl0 = fr.planes[0]
        bpp = pl0["params"].bytes_per_pixel
        row = planes[0].reshape(-1, pl0["out_size"][2])[pl0["out_size"][1] // 2, : 16 * bpp]
        entry["mid_row_first_bytes"] = row.tolist()
        out[name] = entry
        print(name, entry["planes"])
    out["maps"] = {}
    for name, case in MAP_CASES.items():
        fr, kp, pp, mode = map_inputs(case)
        undist = O.stmap_undistort(kp, fr.model, 0, fr.matrices, case["w"], case["h"])
        dist = O.undistort_points(pp, fr.model, 0, fr.rotations, grid=(case["w"], case["h"]), index_mode=mode)
        out["maps"][name] = {"undist": zlib.crc32(undist.tobytes()), "dist": zlib.crc32(dist.tobytes()),
                             "rotations": zlib.crc32(fr.rotations.tobytes()),
                             "dist_centre": [float(v) for v in dist[case["h"] // 2, case["w"] // 2]]}
        print(name, out["maps"][name])
    with open(os.path.join(HERE, "golden.json"), "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
