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


# Host builds (x86-64, build_host below): name -> (ocl_names, bytes per pixel, interpolation, lens model).  The fisheye model with every
# pixel type the reference's OpenCL backend serves (pixel_formats.rs ocl_names; its three-channel types carry a FIXME there: left out) x every sampler (2 bilinear, 4 bicubic, 8 Lanczos4, 10-13 EWA on 16-bit luma); the other
# eight physical lens models on 16-bit luma, bilinear.  Flags stay run-time tests in these builds (assemble(fold_flags=False)).
OCL_NAMES = {
    "luma8": (("uchar", "convert_uchar_sat", "float", "convert_float"), 1), "luma16": (("ushort", "convert_ushort_sat", "float", "convert_float"), 2),
    "rgba8": (("uchar4", "convert_uchar4_sat", "float4", "convert_float4"), 4), "rgba16": (("ushort4", "convert_ushort4_sat", "float4", "convert_float4"), 8),
    "rgbaf": (("float4", "convert_float4", "float4", "convert_float4"), 16), "r32f": (("float", "convert_float", "float", "convert_float"), 4),
    "uv8": (("uchar2", "convert_uchar2_sat", "float2", "convert_float2"), 2), "uv16": (("ushort2", "convert_ushort2_sat", "float2", "convert_float2"), 4),
    "rgbaf16": (("half4", "convert_half4", "float4", "convert_half4_to_float4"), 8),
}
SAMPLERS = {2: "bilinear", 4: "bicubic", 8: "lanczos4", 10: "ewa10", 11: "ewa11", 12: "ewa12", 13: "ewa13"}
HOST_CONFIGS = {}
for _pix, (_names, _bpp) in OCL_NAMES.items():
    for _i in (2, 4, 8):
        HOST_CONFIGS["%s_%s_fisheye" % (_pix, SAMPLERS[_i])] = (_names, _bpp, _i, "opencv_fisheye")
for _i in (10, 11, 12, 13):
    HOST_CONFIGS["luma16_%s_fisheye" % SAMPLERS[_i]] = (OCL_NAMES["luma16"][0], 2, _i, "opencv_fisheye")
for _m in ("opencv_standard", "poly3", "poly5", "ptlens", "insta360", "sony", "generic_polynomial", "gopro"):
    HOST_CONFIGS["luma16_bilinear_" + _m] = (OCL_NAMES["luma16"][0], 2, 2, _m)


DIGITAL = ("gopro_superview", "gopro6_superview", "gopro_hyperview", "gopro_warp", "digital_stretch")
for _d in DIGITAL:      # fisheye under each digital lens (flags & 2 at run time), 16-bit luma and the two chroma layouts of C2 / NV12
    for _pix in ("luma16", "luma8", "uv8"):
        HOST_CONFIGS["%s_bilinear_fisheye+%s" % (_pix, _d)] = (OCL_NAMES[_pix][0], OCL_NAMES[_pix][1], 2, "opencv_fisheye+" + _d)


def digital_functions(digital):
    """What DistortionModel::opencl_functions() returns for a digital lens: the raw string literal of `fn opencl_functions` in
    distortion_models/<digital>.rs (the digital lenses have no .cl file; opencl.rs:186-189 appends this text to the lens model's)."""
    rs = open(os.path.join(REF, "stabilization", "distortion_models", digital + ".rs")).read()
    at = rs.index("fn opencl_functions")
    a = rs.index('r#"', at) + 3
    return rs[a:rs.index('"#', a)]


def assemble(names, bpp, interp, flags, model, fold_flags=True):
    kernel = open(os.path.join(REF, "gpu", "opencl_undistort.cl")).read()
    model, _, digital = model.partition("+")
    lens = open(os.path.join(REF, "stabilization", "distortion_models", model + ".cl")).read()
    if digital:
        lens += digital_functions(digital)
    else:
        lens += ("float2 digital_undistort_point(float2 uv, __global KernelParams *p) { return uv; }\n"
                 "float2 digital_distort_point(float2 uv, __global KernelParams *p) { return uv; }")
    extensions = ""
    if names[1] == "convert_half4":          # opencl.rs:190-197: the fp16 helpers pushed for RGBAf16, a raw string literal of that file
        rs = open(os.path.join(REF, "gpu", "opencl.rs")).read()
        a = rs.index('r#"', rs.index("extensions.push_str(")) + 3
        extensions = rs[a:rs.index('"#', a)]
    kernel = (kernel.replace("LENS_MODEL_FUNCTIONS;", lens).replace("EXTENSIONS;", extensions)
              .replace("DATA_CONVERTF", names[3]).replace("DATA_TYPEF", names[2])
              .replace("DATA_CONVERT", names[1]).replace("DATA_TYPE", names[0])
              .replace("PIXEL_BYTES", str(bpp)).replace("INTERPOLATION", str(interp)))
    for i in range(31 if fold_flags else 0):      # fold_flags=False (host build): `(params->flags & N)` stays a run-time test — the same
        v = 1 << i                                # meaning, and one library then serves every flag combination of a frame
        if v == 4:
            continue
        kernel = kernel.replace("(params->flags & %d)" % v, "true" if (flags & v) == v else "false")
    return kernel


def build_host(name, src, td):
    """The same assembled text compiled for the host cores (x86-64) and linked with oracle/ref_cl_host.c (the OpenCL builtins it
    leaves undefined + the NDRange loop) into oracle/_ref/gfw_ref_cl_<name>.host.so: the second opinion of the CPU side of the suite
    (tests/test_ref_opencl_host.py)."""
    out = os.path.join(OUT, "gfw_ref_cl_%s.host.so" % name)
    cl, obj = os.path.join(td, "h.cl"), os.path.join(td, "h.o")
    open(cl, "w").write(src)
    for cmd in ([CLANG, "-x", "cl", "-cl-std=CL2.0", "-Xclang", "-finclude-default-header", "--target=x86_64-unknown-linux-gnu", "-O2",
                 "-ffp-contract=off", "-fPIC", "-Wno-everything", "-c", cl, "-o", obj],
                [CLANG, "--target=x86_64-unknown-linux-gnu", "-O2", "-ffp-contract=off", "-fPIC", "-shared", "-Wl,-Bsymbolic", "-Wl,-z,defs", "-Wno-everything",
                 os.path.join(HERE, "ref_cl_host.c"), obj, "-lm", "-o", out]):
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("reference OpenCL kernel %s failed to build for the host:\n%s" % (name, r.stderr[-3000:]))
    return out


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
    with tempfile.TemporaryDirectory() as td:
        for name, (names, bpp, interp, model) in HOST_CONFIGS.items():
            built.append(build_host(name, assemble(names, bpp, interp, 0, model, fold_flags=False), td))
            if verbose:
                print("built", built[-1])
    return built


if __name__ == "__main__":
    build(verbose=True)
