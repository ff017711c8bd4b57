"""Explained-residual check against the reference's OWN GPU kernel, for every physical lens model.

oracle/build_ref_cl.py assembles src/core/gpu/opencl_undistort.cl + distortion_models/<model>.cl exactly as OclWrapper::new does
(opencl.rs:181-214) and compiles them offline for gfx950 (oracle/_ref/*.co, built where /root/reference is mounted; only the code objects
travel).  Here the code object runs on the MI355X beside the oracle — and beside libgfwarp itself — on the same inputs.  The twin is NOT
golden (SURVEY.md section 8a lists where the reference's GPU kernels deviate from its CPU path), so the test is not "equal" but
"every difference is one of the documented ones": each pixel on which the two disagree must fall in a class of tests/_refcl.py (a
source coordinate within tau of a 1/32-px bin edge, a rolling-shutter row pick within tau_row of a tie, a negative coordinate under
the twin's rtz rounding, a ray the r-limit test decides) — ZERO unexplained pixels — and the agreement floors sit just under what
MI355X measured (gpurun_out/r03c/ref_residual.log: 99.86-99.90 % identical for opencv_fisheye, 99.62-99.89 % for the other eight
models, whose OpenCL pow / tan / atan stray further from glibc's).  A misreading of the algorithm in the oracle would show up as
wholesale, unexplainable disagreement."""
import numpy as np
import pytest

from gyroflow_amd import abi, synthetic as S, warp
from test_gpu_lens_models import PHYSICAL
from _refcl import classify, oracle_plane, run_reference_cl, smooth

pytestmark = pytest.mark.gpu

TAU_FISHEYE = 2e-4        # px; measured: every difference within 1e-4
TAU_MODELS = 1e-3         # px; the other models' OpenCL builtins (pow, tan, iterative inverses) stray further; measured: all within 1e-3


def explained(fr, name, interp, tau, floor_pct, what):
    dt = np.dtype(abi.PIXEL_TYPES[fr.planes[0]["pixel_type"]][1])
    ref = oracle_plane(fr).view(dt)
    got = run_reference_cl(name, fr.planes[0], fr.matrices).view(dt)
    r = classify(fr, ref, got, interp, taus=(tau,))
    print("%s: %.3f %% identical, %d differ: %s, unexplained %d" % (what, r["identical_pct"], r["differ"], r.get("classes"), r.get("unexplained", -1)))
    assert r["identical_pct"] >= floor_pct, r
    assert "classes" in r, r
    assert r["unexplained"] == 0, r["unexplained_examples"]
    if fr.planes[0]["params"].r_limit <= 0.0:
        assert r["classes"]["invalid"] == 0, r          # without an r-limit the twins must agree on which rays are valid
    return ref, got, r


@pytest.mark.parametrize("name,fmt,interp", [("luma16_bilinear_fisheye", "YUV422P16LE", 2), ("luma8_bilinear_fisheye", "NV12", 2),
                                            ("luma16_lanczos4_fisheye", "YUV422P16LE", 8), ("rgbaf_bilinear_fisheye", "RGBAF32", 2)])
def test_every_difference_from_the_reference_kernel_is_a_documented_one(name, fmt, interp):
    fr = S.SyntheticFrame(fmt, 640, 360, seed=0x9F10 + 3, interpolation=interp)
    explained(fr, name, interp, TAU_FISHEYE, 99.8, name)
    if fmt == "RGBAF32":
        return
    # a smooth frame: a one-bin coordinate difference moves the value by a few codes at most
    smooth(fr)
    dt = np.dtype(abi.PIXEL_TYPES[fr.planes[0]["pixel_type"]][1])
    ref = oracle_plane(fr).view(dt).astype(np.int64)
    got = run_reference_cl(name, fr.planes[0], fr.matrices).view(dt).astype(np.int64)
    d = np.abs(ref - got)
    print("%s smooth frame: %.3f %% identical, max |difference| %d code values" % (name, 100.0 * float(np.mean(d == 0)), int(d.max())))
    assert d.max() <= (4 if dt.itemsize == 2 else 1), int(d.max())


@pytest.mark.parametrize("model", sorted(m for m in PHYSICAL if m != "opencv_fisheye"))
def test_every_lens_model_against_the_reference_kernel(model):
    w, h = 640, 360
    lens = S.gopro_style_lens(w, h)
    lens["model"] = model
    lens["k"] = PHYSICAL[model] + [0.0] * (12 - len(PHYSICAL[model]))
    if model == "gopro":
        lens["r_limit"] = 2.5
    fr = S.SyntheticFrame("YUV422P16LE", w, h, seed=0x9F10 + 7, lens=lens, fov=1.2)
    explained(fr, "luma16_bilinear_" + model, 2, TAU_MODELS, 99.5, model)


def test_libgfwarp_itself_against_the_reference_kernel_c2_1080p():
    """The product (not the oracle) beside the reference's kernel: C2's configuration at 1920x1080, luma plane."""
    fr = S.SyntheticFrame("YUV422P16LE", 1920, 1080, seed=0x9F10 + 11)
    got_lib = warp.run_frame(fr)[0].view(np.uint16)
    assert warp.last_backend().startswith("yuv_fused_p1")
    ref_cl = run_reference_cl("luma16_bilinear_fisheye", fr.planes[0], fr.matrices).view(np.uint16)
    r = classify(fr, got_lib, ref_cl, 2, taus=(TAU_FISHEYE,))
    print("libgfwarp vs reference OpenCL kernel, 1920x1080: %.3f %% identical, %d differ: %s" % (r["identical_pct"], r["differ"], r["classes"]))
    # at this width a coordinate's ulp is 1.2e-4 px: more pixels sit within a few ulp of a bin edge than at 640x360 (measured 99.39 %)
    assert r["identical_pct"] >= 99.2 and r["unexplained"] == 0 and r["classes"]["invalid"] == 0, r
    assert np.array_equal(got_lib, oracle_plane(fr).view(np.uint16))
