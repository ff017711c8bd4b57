#!/usr/bin/env python3
"""Generate tests/golden/ref_golden.json: CRC32 of every output plane produced by THE REFERENCE'S OWN warp kernel.

The reference ships no golden vectors for this path and its Rust cannot be built here — but its OpenCL twin
of the path can: oracle/build_ref_cl.py assembles src/core/gpu/opencl_undistort.cl + distortion_models/<model>.cl from /root/reference
exactly as OclWrapper::new does (opencl.rs:181-214) and compiles that text for the host cores (x86-64), oracle/ref_cl_host.c supplies
the OpenCL builtins (the transcendental ones from glibc, which is what the reference's CPU path calls) and the NDRange loop.  This
script runs that build — the reference's code, not a restatement — on seeded synthetic frames and freezes what it wrote.

The twin is the reference's GPU backend; it deviates from the reference's CPU path in a handful of documented places (SURVEY.md
section 8a: negative coordinates under convert_int_sat_rtz(0.5 + x), .cl:355; the r-limit formula, .cl:402; the feather zone of
background mode 3, .cl:620-622; NaN coordinates; the colour-range fix, .cl:157-160; EWA).  The cases below are configurations in
which none of those can fire (every source coordinate inside the frame or clamped into it), so the two backends of the reference
must agree there and the fixture pins the CPU path too: tests/test_ref_golden.py holds the oracle (CPU tier) and libgfwarp (GPU
tier) to these numbers bit for bit.  Configurations where the deviations do fire are covered by the explained-residual tests of
tests/test_ref_opencl_host.py instead.

Needs /root/reference (for the build); run from the repo root:  python tests/golden/make_ref_golden.py
"""
import json
import os
import sys
import zlib

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from gyroflow_amd import abi, synthetic as S  # noqa: E402

# pixel type -> host build family (oracle/build_ref_cl.py OCL_NAMES; BGRA8 / AYUV16 share the four-channel kernels, as in pixel_formats.rs)
PIX = {"Luma8": "luma8", "Luma16": "luma16", "RGBA8": "rgba8", "BGRA8": "rgba8", "RGBA16": "rgba16", "AYUV16": "rgba16",
       "RGBAf": "rgbaf", "RGBAf16": "rgbaf16", "R32f": "r32f", "UV8": "uv8", "UV16": "uv16"}
SAMPLERS = {2: "bilinear", 4: "bicubic", 8: "lanczos4"}
MODEL_TAG = {"opencv_fisheye": "fisheye"}

# lens coefficients of the non-fisheye cases (the values tests/test_gpu_lens_models.py uses)
LENS_K = {
    "generic_polynomial": [1.0, 0.01, -0.05, 0.02, 0.003, -0.001, 0.0005],
    "gopro": [0.0, 1.0, 0.01, -0.12, 0.02, 0.01, -0.004],
    "ptlens": [0.01, -0.03, 0.02],
    "sony": [1.0, 0.01, -0.05, 0.02, 0.003, -0.001],
}

CASES = {
    # BASELINE.json configurations at full size
    "c2_yuv422p16_3840x2160_rs": dict(fmt="YUV422P16LE", w=3840, h=2160, seed=0x9F10),
    "c1_nv12_1920x1080_constquat": dict(fmt="NV12", w=1920, h=1080, seed=0x9F10, readout_ms=0.0, quat=(5.0, 2.0, 3.0)),
    "c4_rgbaf32_crop_1280x720": dict(fmt="RGBAF32", w=1280, h=720, seed=21, fov=0.82, ov={"translation2d": (13.25, -7.5)}),
    "c4_gbrapf32_crop_640x360": dict(fmt="GBRAPF32LE", w=640, h=360, seed=21, fov=0.82, ov={"translation2d": (13.25, -7.5)}),
    # the golden.json cases the twin can pin
    "c2_yuv422p16_480x270_rs": dict(fmt="YUV422P16LE", w=480, h=270, seed=0x9F10),
    "p010_320x192_rs": dict(fmt="P010", w=320, h=192, seed=11),
    "yuv420p10_320x192_rs": dict(fmt="YUV420P10LE", w=320, h=192, seed=11),
    "yuv422p16_lanczos4": dict(fmt="YUV422P16LE", w=192, h=128, seed=5, interp=8),
    "yuv422p16_bicubic": dict(fmt="YUV422P16LE", w=192, h=128, seed=5, interp=4),
    "yuv422p16_mirror": dict(fmt="YUV422P16LE", w=256, h=160, seed=9, fov=2.5, ov={"background_mode": 2}),
    "nv12_horizontal_rs": dict(fmt="NV12", w=256, h=160, seed=4, hrs=True),
    # formats x samplers
    "nv12_lanczos4_640x360": dict(fmt="NV12", w=640, h=360, seed=0x1234, interp=8),
    "nv12_bicubic_640x360": dict(fmt="NV12", w=640, h=360, seed=0x1234, interp=4),
    "p210_bilinear_640x360": dict(fmt="P210LE", w=640, h=360, seed=0x1234),
    "p010_lanczos4_640x360": dict(fmt="P010LE", w=640, h=360, seed=0x1234, interp=8),
    "yuv420p_bilinear_642x362": dict(fmt="YUV420P", w=642, h=362, seed=0x1236),
    "yuv444p16_bicubic_640x360": dict(fmt="YUV444P16LE", w=640, h=360, seed=0x1234, interp=4),
    "ayuv64_bilinear_640x360": dict(fmt="AYUV64LE", w=640, h=360, seed=0x1234),
    "rgba_lanczos4_640x360": dict(fmt="RGBA", w=640, h=360, seed=0x1234, interp=8),
    "bgra_bilinear_640x360": dict(fmt="BGRA", w=640, h=360, seed=0x1234),
    "rgba64_bicubic_640x360": dict(fmt="RGBA64BE", w=640, h=360, seed=0x1234, interp=4),
    "rgbaf32_lanczos4_640x360": dict(fmt="RGBAF32", w=640, h=360, seed=0x1234, interp=8),
    "rgbaf16_bilinear_640x360": dict(fmt="RGBAF16", w=640, h=360, seed=0x1234),
    "rgbaf16_lanczos4_640x360": dict(fmt="RGBAF16", w=640, h=360, seed=0x1234, interp=8),
    "gbrapf32_bicubic_640x360": dict(fmt="GBRAPF32LE", w=640, h=360, seed=0x1234, interp=4),
    # geometry
    "yuv422p16_fov05_hrs": dict(fmt="YUV422P16LE", w=640, h=360, seed=77, fov=0.5, hrs=True, readout_ms=25.0),
    "yuv422p16_edge_repeat": dict(fmt="YUV422P16LE", w=640, h=360, seed=78, fov=1.8, ov={"background_mode": 1}),
    "yuv422p16_stretch": dict(fmt="YUV422P16LE", w=640, h=360, seed=83, ov={"input_vertical_stretch": 1.1, "input_horizontal_stretch": 0.9}),
    "yuv422p16_fill_background": dict(fmt="YUV422P16LE", w=640, h=360, seed=85, fov=1.4, flags=4, bg=(0.3, 0.6, 0.9, 1.0)),
    "yuv422p16_644x362_to_512x300": dict(fmt="YUV422P16LE", w=644, h=362, seed=88, out=(512, 300)),
    "yuv422p16_1920x1080_rs": dict(fmt="YUV422P16LE", w=1920, h=1080, seed=87),
    # other lens models, field of view inside the frame
    "generic_polynomial_640x360": dict(fmt="YUV422P16LE", w=640, h=360, seed=0x9F17, model="generic_polynomial"),
    "gopro_640x360": dict(fmt="YUV422P16LE", w=640, h=360, seed=0x9F17, model="gopro", r_limit=2.5),
    "ptlens_640x360": dict(fmt="YUV422P16LE", w=640, h=360, seed=0x9F17, model="ptlens"),
    "sony_640x360": dict(fmt="YUV422P16LE", w=640, h=360, seed=0x9F17, model="sony"),
    # digital lenses on top of the fisheye (flags & 2), with and without the lens-correction blend
    "superview_640x360": dict(fmt="YUV422P16LE", w=640, h=360, seed=37, fov=0.8, digital="gopro_superview"),
    "superview_nv12_lca06_640x360": dict(fmt="NV12", w=640, h=360, seed=37, fov=1.0, digital="gopro_superview", ov={"lens_correction_amount": 0.6}),
    "superview6_640x360": dict(fmt="YUV422P16LE", w=640, h=360, seed=37, fov=0.8, digital="gopro6_superview"),
    "hyperview_lca06_640x360": dict(fmt="YUV422P16LE", w=640, h=360, seed=37, fov=0.8, digital="gopro_hyperview", ov={"lens_correction_amount": 0.6}),
    "digital_stretch_640x360": dict(fmt="YUV422P16LE", w=640, h=360, seed=37, digital="digital_stretch", ov={"digital_lens_params": [1.1, 0.95]}),
    # IBIS / OIS terms in the matrices (flags & 256, as get_kernel_flags sets it: mod.rs:226-251), input rotation by quarter turns
    "ibis_terms_640x360": dict(fmt="YUV422P16LE", w=640, h=360, seed=53, ibis=True),
    "input_rotation_90_nv12": dict(fmt="NV12", w=640, h=360, seed=43, rot=90.0),
    "input_rotation_180_640x360": dict(fmt="YUV422P16LE", w=640, h=360, seed=43, rot=180.0),
    "gopro_warp_640x360": dict(fmt="YUV422P16LE", w=640, h=360, seed=37, digital="gopro_warp",
                               ov={"digital_lens_params": [1.32, -1.2, 1.6, -0.4, 0.1, 0.0, 0.0, -0.1, 0.95, 0.4, -0.7, -0.35, 1.1, 0.35, 1.3333334]}),
}


def build(case):
    q = S.quat_from_euler_deg(*case["quat"]) if "quat" in case else None
    lens = None
    if "model" in case or "digital" in case:
        lens = S.gopro_style_lens(case["w"], case["h"])
    if "model" in case:
        lens["model"] = case["model"]
        lens["k"] = LENS_K[case["model"]] + [0.0] * (12 - len(LENS_K[case["model"]]))
        lens["r_limit"] = case.get("r_limit", 0.0)
    if "digital" in case:
        lens["digital"] = case["digital"]
    flags = case.get("flags", 0) | (abi.FLAG_HAS_IBIS_DATA if case.get("ibis") else 0)
    fr = S.SyntheticFrame(case["fmt"], case["w"], case["h"], seed=case["seed"], fov=case.get("fov", 1.0),
                          readout_ms=case.get("readout_ms", 16.0), interpolation=case.get("interp", 2), constant_quat=q,
                          horizontal_rs=case.get("hrs", False), out_size=case.get("out"), lens=lens, flags=flags,
                          background_rgba=case.get("bg", (0.0, 0.0, 0.0, 0.0)), base_overrides=case.get("ov"))
    if case.get("ibis"):                        # shift, roll and offset per row; every seventh row without data
        y = np.arange(fr.matrices.shape[0], dtype=np.float32)
        fr.matrices[:, 9] = 1.5 * np.sin(y * 0.05)
        fr.matrices[:, 10] = -0.8 * np.cos(y * 0.03)
        fr.matrices[:, 11] = 0.004 * np.sin(y * 0.02)
        fr.matrices[:, 12] = 0.6
        fr.matrices[:, 13] = -0.4
        fr.matrices[::7, 9:14] = 0.0
    if "rot" in case:
        for pl in fr.planes:
            pl["params"].input_rotation = case["rot"]
    return fr


def host_config(fr, plane, case):
    model = fr.lens["model"]
    digital = ("+" + case["digital"]) if "digital" in case else ""
    return "%s_%s_%s%s" % (PIX[plane["pixel_type"]], SAMPLERS[case.get("interp", 2)], MODEL_TAG.get(model, model), digital)


def run_reference(fr, case):
    """Every plane of the frame through the reference's own kernel (one NDRange per plane, as rendering/mod.rs:542 issues them)."""
    from _refcl import run_reference_cl_host
    return [run_reference_cl_host(host_config(fr, pl, case), pl, fr.matrices) for pl in fr.planes]


def main():
    out = {}
    for name, case in CASES.items():
        fr = build(case)
        planes = run_reference(fr, case)
        entry = {"planes": [zlib.crc32(p.tobytes()) for p in planes],
                 "src": [zlib.crc32(pl["src"].tobytes()) for pl in fr.planes],
                 "matrices": zlib.crc32(fr.matrices.tobytes()),
                 "kernels": [host_config(fr, pl, case) for pl in fr.planes]}
        pl0 = fr.planes[0]
        bpp = pl0["params"].bytes_per_pixel
        entry["mid_row_first_bytes"] = planes[0].reshape(-1, pl0["out_size"][2])[pl0["out_size"][1] // 2, : 16 * bpp].tolist()
        out[name] = entry
        print(name, entry["planes"])
    with open(os.path.join(HERE, "ref_golden.json"), "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
