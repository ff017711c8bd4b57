#!/usr/bin/env python3
"""Calibration run for tests/test_gpu_ref_opencl.py (GPU box): where do the reference's own OpenCL kernel and the oracle differ, and why?

For every configuration the reference kernel (oracle/_ref/*.co) and the oracle warp the same noisy frame; every pixel on which they
differ is classified from the ORACLE's coordinates:
  bin   the source coordinate (x or y) lies within TAU px of a 1/32-px bin edge (OpenCL's atan / native divide differ from glibc's
        by a few ulp, and the GPU twin rounds with convert_int_sat_rtz(0.5 + x): SURVEY.md section 8a)
  row   the first-pass coordinate that picks the rolling-shutter row lies within TAU_ROW of a half-integer (the neighbouring row's
        matrix moves the sample by up to a few 1/32 px)
  neg   a coordinate is negative: the twin rounds with convert_int_sat_rtz(0.5 + x) (.cl:355), which for x < 0 lands one 1/32-px bin
        above Rust's round-half-away
  edge  (otherwise) a tap of the sample falls outside the source rect (the twins treat the border differently)
  none  unexplained
Prints one line per configuration and the unexplained pixels; writes gpurun_out/ref_residual.json."""
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
from test_gpu_ref_opencl import oracle_plane, run_reference_cl  # noqa: E402
from test_gpu_lens_models import PHYSICAL  # noqa: E402


def coords_of(fr, pts):
    """oracle (ok, u, v) and first-pass row coordinate of output pixels pts (list of (x, y)) of plane 0"""
    pl = fr.planes[0]
    p = pl["params"]
    res = []
    mid = np.ascontiguousarray(fr.matrices[p.matrix_count // 2: p.matrix_count // 2 + 1], dtype=np.float32)
    p1 = abi.KernelParams.from_buffer_copy(bytes(p))
    p1.matrix_count = 1
    for (x, y) in pts:
        ok, u, v = O.undistort_coord(p, fr.model, fr.digital, fr.matrices, float(x), float(y))
        ok1, u1, v1 = O.undistort_coord(p1, fr.model, fr.digital, mid, float(x), float(y))
        res.append((ok, u, v, ok1, u1, v1))
    return res


def classify(fr, ref, got, interp, taus=(5e-5, 1e-4, 2e-4, 4e-4, 1e-3), tau_row=2e-3):
    pl = fr.planes[0]
    w, h = pl["out_size"][0], pl["out_size"][1]
    sw, sh = pl["size"][0], pl["size"][1]
    n = abi.PIXEL_TYPES[pl["pixel_type"]][2]
    a = ref.reshape(h, -1)[:, :w * n].reshape(h, w, n)
    b = got.reshape(h, -1)[:, :w * n].reshape(h, w, n)
    diff = np.any(a != b, axis=2)
    ys, xs = np.nonzero(diff)
    out = {"pixels": int(w * h), "differ": int(len(xs)), "identical_pct": 100.0 * (1.0 - len(xs) / float(w * h))}
    if a.dtype.kind in "ui":
        out["max_abs_diff"] = int(np.max(np.abs(a.astype(np.int64) - b.astype(np.int64)))) if len(xs) else 0
    else:
        with np.errstate(all="ignore"):
            out["max_abs_diff"] = float(np.nanmax(np.abs(a.astype(np.float64) - b.astype(np.float64)))) if len(xs) else 0.0
    if len(xs) > 6000:
        out["note"] = "too many differing pixels to classify"
        return out
    cs = coords_of(fr, list(zip(xs.tolist(), ys.tolist())))
    off = {2: 0.0, 4: 1.0, 8: 3.0}[interp]
    cls = {"row": 0, "neg": 0, "edge": 0}
    for t in taus:
        cls["bin@%g" % t] = 0
    unexplained, edge_examples = [], []
    for (x, y), (ok, u, v, ok1, u1, v1) in zip(zip(xs.tolist(), ys.tolist()), cs):
        def edge_dist(c):
            t = (np.float32(c) - np.float32(off)) * np.float32(32.0)
            fr_ = float(t) - np.floor(float(t))
            return abs(fr_ - 0.5) / 32.0
        d = min(edge_dist(u), edge_dist(v)) if ok else 0.0
        p = pl["params"]
        hrs = bool(p.flags & 16)
        pv = u1 if hrs else v1
        drow = abs((pv - np.floor(pv)) - 0.5) if ok1 else 0.0
        sx, sy = int(np.floor(u - off)), int(np.floor(v - off))
        is_edge = (not ok) or sx < 0 or sy < 0 or sx + interp > sw or sy + interp > sh
        hit = False
        for t in taus:
            if d <= t:
                cls["bin@%g" % t] += 1
                hit = True
        if not hit:
            if drow <= tau_row and p.matrix_count > 1:
                cls["row"] += 1
            elif ok and ((u - off) < 0.0 or (v - off) < 0.0):
                cls["neg"] += 1
            elif is_edge:
                cls["edge"] += 1
                if len(edge_examples) < 8:
                    edge_examples.append({"x": x, "y": y, "ok": bool(ok), "u": u, "v": v, "ref": a[y, x].tolist(), "got": b[y, x].tolist()})
            else:
                unexplained.append({"x": x, "y": y, "u": u, "v": v, "bin_dist": float(d), "row_dist": float(drow), "ref": a[y, x].tolist(), "got": b[y, x].tolist()})
    out["classes"] = cls
    out["unexplained"] = len(unexplained)
    out["unexplained_examples"] = unexplained[:12]
    out["edge_examples"] = edge_examples
    return out


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
        print(name, json.dumps({k: v for k, v in r.items() if k not in ("unexplained_examples", "edge_examples")}))
        for ex in r.get("edge_examples", []):
            print("    edge", ex)
        for ex in r.get("unexplained_examples", []):
            print("    unexplained", ex)
        sys.stdout.flush()
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    json.dump(results, open(os.path.join(ROOT, "gpurun_out", "ref_residual.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
