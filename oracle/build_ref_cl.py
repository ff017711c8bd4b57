#!/usr/bin/env python3
"""Build the reference's OWN OpenCL warp kernel for gfx950 as a tolerance-level second opinion (test infrastructure).

The reference's GPU twin of the path (src/core/gpu/opencl_undistort.cl + distortion_models/<model>.cl) is assembled exactly
the way OclWrapper::new does it (src/core/gpu/opencl.rs:181-214: LENS_MODEL_FUNCTIONS / DATA_TYPE* / PIXEL_BYTES /
INTERPOLATION substitutions, every `(params->flags & N)` except 4 folded to true/false) from the sources where they lie
under /root/reference, and compiled offline with the ROCm clang into oracle/_ref/gfw_ref_cl_<name>.co.  Only the built code
objects are kept (git-ignored; they travel to the GPU box); no reference source enters the repository.

It is NOT golden: SURVEY.md section 8a lists where the reference's GPU kernels deviate from its CPU path (sub-pixel rounding
`convert_int_sat_rtz(0.5+x)`, the r-limit test, fisheye clamps, range-fix order).  tests/test_gpu_ref_opencl.py expects
>= 99.9 % identical pixels and <= 1 LSB elsewhere — enough to catch a misreading of the algorithm, which is its purpose.
"""
import os
import subprocess
import sys
import tempfile

REF = "/root/reference/src/core"
HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(HERE, "_ref")
CLANG = "/opt/rocm/lib/llvm/bin/clang"

# name -> (ocl_names (pixel_formats.rs), bytes_per_pixel, interpolation, flags, lens model)
CONFIGS = {
    "luma16_bilinear_fisheye": (("ushort", "convert_ushort_sat", "float", "convert_float"), 2, 2, 0, "opencv_fisheye"),
    "luma8_bilinear_fisheye": (("uchar", "convert_uchar_sat", "float", "convert_float"), 1, 2, 0, "opencv_fisheye"),
    "luma16_lanczos4_fisheye": (("ushort", "convert_ushort_sat", "float", "convert_float"), 2, 8, 0, "opencv_fisheye"),
    "rgbaf_bilinear_fisheye": (("float4", "convert_float4", "float4", "convert_float4"), 16, 2, 0, "opencv_fisheye"),
}
# the other physical lens models, 16-bit luma, bilinear (tests/test_staged_ref_opencl_models.py: staged until their agreement
# thresholds have been calibrated on the device)
for _m in ("opencv_standard", "poly3", "poly5", "ptlens", "insta360", "sony", "generic_polynomial", "gopro"):
    CONFIGS["luma16_bilinear_" + _m] = (("ushort", "convert_ushort_sat", "float", "convert_float"), 2, 2, 0, _m)


def assemble(names, bpp, interp, flags, model):
    kernel = open(os.path.join(REF, "gpu", "opencl_undistort.cl")).read()
    lens = open(os.path.join(REF, "stabilization", "distortion_models", model + ".cl")).read()
    lens += ("float2 digital_undistort_point(float2 uv, __global KernelParams *p) { return uv; }\n"
             "float2 digital_distort_point(float2 uv, __global KernelParams *p) { return uv; }")
    kernel = (kernel.replace("LENS_MODEL_FUNCTIONS;", lens).replace("EXTENSIONS;", "")
              .replace("DATA_CONVERTF", names[3]).replace("DATA_TYPEF", names[2])
              .replace("DATA_CONVERT", names[1]).replace("DATA_TYPE", names[0])
              .replace("PIXEL_BYTES", str(bpp)).replace("INTERPOLATION", str(interp)))
    for i in range(31):
        v = 1 << i
        if v == 4:
            continue
        kernel = kernel.replace("(params->flags & %d)" % v, "true" if (flags & v) == v else "false")
    return kernel


def build(verbose=False):
    if not os.path.isdir(REF):
        return []
    os.makedirs(OUT, exist_ok=True)
    built = []
    for name, (names, bpp, interp, flags, model) in CONFIGS.items():
        out = os.path.join(OUT, "gfw_ref_cl_%s.co" % name)
        src = assemble(names, bpp, interp, flags, model)
        with tempfile.TemporaryDirectory() as td:                      # the assembled source never lands in the repository
            cl = os.path.join(td, "k.cl")
            open(cl, "w").write(src)
            cmd = [CLANG, "-x", "cl", "-cl-std=CL2.0", "-Xclang", "-finclude-default-header", "--target=amdgcn-amd-amdhsa", "-mcpu=gfx950",
                   "-O2", "-ffp-contract=off", "-mcode-object-version=5", "-Wno-everything", cl, "-o", out]
            r = subprocess.run(cmd, capture_output=True, text=True)
            if r.returncode != 0:
                raise RuntimeError("reference OpenCL kernel %s failed to build:\n%s" % (name, r.stderr[-3000:]))
        built.append(out)
        if verbose:
            print("built", out)
    return built


if __name__ == "__main__":
    build(verbose=True)
