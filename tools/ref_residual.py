#!/usr/bin/env python3
"""Calibration run for tests/test_gpu_ref_opencl.py (GPU box): where do the reference's own OpenCL kernel and the oracle differ, and why?
Runs every configuration of oracle/build_ref_cl.py on a noisy frame, classifies the differing pixels (tests/_refcl.py) at several bin-edge
tolerances, prints one line per configuration and the unexplained pixels; writes gpurun_out/ref_residual.json."""
import ctypes as C
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from gyroflow_amd import abi, synthetic as S  # noqa: E402
import _oracle as O  # noqa: E402
from _refcl import classify, oracle_plane, run_reference_cl  # noqa: E402
from test_gpu_lens_models import PHYSICAL  # noqa: E402


def main():
    results = {}
    w, h = 640, 360
    cases = [("luma16_bilinear_fisheye", "YUV422P16LE", 2, None), ("luma8_bilinear_fisheye", "NV12", 2, None),
             ("luma16_lanczos4_fisheye", "YUV422P16LE", 8, None), ("rgbaf_bilinear_fisheye", "RGBAF32", 2, None)]
    for m in sorted(PHYSICAL):
        if m != "opencv_fisheye":
            cases.append(("luma16_bilinear_" + m, "YUV422P16LE", 2, m))
    for name, fmt, interp, model in cases:
        kw = {}
        if model:
            lens = S.gopro_style_lens(w, h)
            lens["model"] = model
            lens["k"] = PHYSICAL[model] + [0.0] * (12 - len(PHYSICAL[model]))
            if model == "gopro":
                lens["r_limit"] = 2.5
            kw = {"lens": lens, "fov": 1.2}
        try:
            fr = S.SyntheticFrame(fmt, w, h, seed=0x9F10 + (7 if model else 3), interpolation=interp, **kw)
            dt = np.dtype(abi.PIXEL_TYPES[fr.planes[0]["pixel_type"]][1])
            ref = oracle_plane(fr).view(dt)
            got = run_reference_cl(name, fr.planes[0], fr.matrices).view(dt)
            r = classify(fr, ref, got, interp)
        except BaseException as e:      # pytest.skip raises a BaseException subclass
            r = {"error": repr(e)}
        results[name] = r
        print(name, json.dumps({k: v for k, v in r.items() if k != "unexplained_examples"}))
        for ex in r.get("unexplained_examples", []):
            print("    unexplained", ex)
        sys.stdout.flush()
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    json.dump(results, open(os.path.join(ROOT, "gpurun_out", "ref_residual.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
