"""STAGED — the reference's own OpenCL kernel as a tolerance-level second opinion for EVERY physical lens model.

tests/test_gpu_ref_opencl.py runs opencl_undistort.cl + opencv_fisheye.cl beside the oracle (99.86 % identical pixels).  Here the
same is done with the other eight distortion_models/*.cl (code objects from oracle/build_ref_cl.py).  The agreement levels of
these models are not known yet (OpenCL's pow / atan / tan are not glibc's, some GPU twins iterate differently), so the
thresholds below are deliberately loose and the file runs only with `-m gpu_staged`; once measured on the device the numbers go
into the assertions and the tests join `-m gpu`.  What it can already exclude is a misreading of a lens model in the oracle:
that shows up as wholesale disagreement, not as a fraction of a percent."""
import numpy as np
import pytest

from gyroflow_amd import abi, synthetic as S
from test_gpu_lens_models import PHYSICAL
from test_gpu_ref_opencl import oracle_plane, run_reference_cl, smooth

pytestmark = pytest.mark.gpu_staged


@pytest.mark.parametrize("model", sorted(m for m in PHYSICAL if m != "opencv_fisheye"))
def test_reference_opencl_lens_model_agrees_with_the_oracle(model):
    w, h = 640, 360
    lens = S.gopro_style_lens(w, h)
    lens["model"] = model
    lens["k"] = PHYSICAL[model] + [0.0] * (12 - len(PHYSICAL[model]))
    if model == "gopro":
        lens["r_limit"] = 2.5
    fr = S.SyntheticFrame("YUV422P16LE", w, h, seed=0x9F10 + 7, lens=lens, fov=1.2)
    name = "luma16_bilinear_" + model
    ref = oracle_plane(fr).view(np.uint16)
    got = run_reference_cl(name, fr.planes[0], fr.matrices).view(np.uint16)
    same = float(np.mean(ref == got))
    print("%s noisy frame: %.3f %% of the pixels identical" % (name, 100.0 * same))
    assert same >= 0.90, same
    smooth(fr)
    ref = oracle_plane(fr).view(np.uint16).astype(np.int64)
    got = run_reference_cl(name, fr.planes[0], fr.matrices).view(np.uint16).astype(np.int64)
    d = np.abs(ref - got)
    print("%s smooth frame: %.3f %% identical, max |difference| %d code values" % (name, 100.0 * float(np.mean(d == 0)), int(d.max())))
    assert np.percentile(d, 99.9) <= 16, float(np.percentile(d, 99.9))
