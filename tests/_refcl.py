"""Shared helpers of the reference-OpenCL second opinion (tests/test_gpu_ref_opencl.py, tools/ref_residual.py): run the reference's own
kernel (oracle/_ref/*.co, built by oracle/build_ref_cl.py from /root/reference) through the HIP module API, and classify every pixel on
which it differs from the oracle by WHY the reference's GPU twin may differ from its CPU path there:
  bin   the source coordinate (x or y) lies within max(tau, 4 ulp, 1e-6 of its distance from the principal point) px of a 1/32-px bin edge — OpenCL's atan / pow / native divide are not glibc's,
        and the twin rounds the sub-pixel index with convert_int_sat_rtz(0.5 + x) (.cl:355), so a coordinate a few ulp away lands in
        the neighbouring bin
  row   the first-pass coordinate that picks the rolling-shutter row lies within tau_row of a half-integer: the neighbouring row's
        matrix moves the sample by a few 1/32 px
  neg   a coordinate is negative: for x < 0 the twin's rtz rounding lands one bin above Rust's round-half-away
  invalid  the oracle rejects the ray (w <= 0 or the r-limit test, whose formula differs between the twins: cpu_undistort.rs:139, .cl:402)
Anything else is unexplained."""
import ctypes as C
import os

import numpy as np
import pytest

from gyroflow_amd import abi
import _oracle as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run_reference_cl(name, pl, matrices, block=(64, 4)):
    import torch
    path = os.path.join(ROOT, "oracle", "_ref", "gfw_ref_cl_%s.co" % name)
    if not os.path.exists(path):
        pytest.skip("reference OpenCL code object not built (needs /root/reference at build time)")
    hip = C.CDLL("libamdhip64.so")
    dev = torch.device("cuda", 0)
    torch.cuda.synchronize()
    mod, fn = C.c_void_p(), C.c_void_p()
    assert hip.hipModuleLoad(C.byref(mod), path.encode()) == 0
    assert hip.hipModuleGetFunction(C.byref(fn), mod, b"undistort_image") == 0
    src = torch.from_numpy(pl["src"]).to(dev)
    dst = torch.from_numpy(pl["dst"].copy()).to(dev)
    prm = torch.frombuffer(bytearray(bytes(pl["params"])), dtype=torch.uint8).to(dev)
    mat = torch.from_numpy(np.ascontiguousarray(matrices, dtype=np.float32)).to(dev)
    drawing = torch.zeros(16, dtype=torch.uint8, device=dev)
    mesh = torch.zeros(16, dtype=torch.float32, device=dev)
    ptrs = [C.c_void_p(t.data_ptr()) for t in (src, dst, prm, mat, drawing, mesh)]
    args = (C.c_void_p * 6)(*[C.cast(C.byref(p), C.c_void_p) for p in ptrs])
    ow, oh = pl["out_size"][0], pl["out_size"][1]
    assert ow % block[0] == 0 and oh % block[1] == 0
    hip.hipModuleLaunchKernel.argtypes = [C.c_void_p] + [C.c_uint] * 6 + [C.c_uint, C.c_void_p, C.c_void_p, C.c_void_p]
    rc = hip.hipModuleLaunchKernel(fn, ow // block[0], oh // block[1], 1, block[0], block[1], 1, 0, None, args, None)
    assert rc == 0, rc
    assert hip.hipDeviceSynchronize() == 0
    out = dst.cpu().numpy()
    hip.hipModuleUnload(mod)
    return out


def oracle_plane(fr, idx=0):
    pl = fr.planes[idx]
    dst = pl["dst"].copy()
    assert O.undistort_image(pl["src"], pl["size"], dst, pl["out_size"], pl["params"], pl["pixel_type"], fr.model, fr.digital, fr.matrices) == 1
    return dst


def smooth(fr, idx=0):
    """Replace the plane's noise by a smooth ramp (so that a one-bin coordinate difference moves the value by ~1 code)."""
    pl = fr.planes[idx]
    w, h, stride = pl["size"]
    dt = np.dtype(abi.PIXEL_TYPES[pl["pixel_type"]][1])
    yy, xx = np.meshgrid(np.arange(h), np.arange(w), indexing="ij")
    top = 60000.0 if dt.itemsize == 2 else 250.0
    img = ((xx / (w - 1) * 0.6 + yy / (h - 1) * 0.4) * top).astype(dt)
    view = pl["src"].reshape(h, stride)[:, :w * dt.itemsize]
    view[:] = img.view(np.uint8).reshape(h, w * dt.itemsize)



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
    """-> dict: differing pixels by class ("bin@tau" cumulative for every tau in taus; a pixel counts as explained by a bin edge at the
    largest tau), "row", "neg", and the unexplained ones with their coordinates"""
    pl = fr.planes[0]
    w, h = pl["out_size"][0], pl["out_size"][1]
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
    if len(xs) > 40000:
        out["note"] = "too many differing pixels to classify"
        return out
    cs = coords_of(fr, list(zip(xs.tolist(), ys.tolist())))
    off = {2: 0.0, 4: 1.0, 8: 3.0}[interp]
    cls = {"row": 0, "neg": 0, "invalid": 0}
    for t in taus:
        cls["bin@%g" % t] = 0
    unexplained = []
    for (x, y), (ok, u, v, ok1, u1, v1) in zip(zip(xs.tolist(), ys.tolist()), cs):
        def edge_dist(c, centre):
            """distance of coordinate c to the nearest 1/32-px bin edge, less what the coordinate's magnitude adds to the tolerance's floor:
            the tolerance is max(tau, 4 ulp(c), 1e-6 |c - centre|) — the twins' builtins (OpenCL atan / sqrt / divide against glibc's) differ
            by a few ulp RELATIVE to the ray, i.e. in proportion to the distance from the principal point, and a coordinate beyond 1024
            has an ulp of 1.2e-4 px by itself"""
            t = (np.float32(c) - np.float32(off)) * np.float32(32.0)
            fr_ = float(t) - np.floor(float(t))
            extra = max(4.0 * float(np.spacing(np.float32(abs(c)))), 1e-6 * abs(float(c) - centre))
            return abs(fr_ - 0.5) / 32.0 - max(0.0, extra - min(taus))
        d = min(edge_dist(u, float(pl["params"].c[0])), edge_dist(v, float(pl["params"].c[1])))
        p = pl["params"]
        hrs = bool(p.flags & 16)
        pv = u1 if hrs else v1
        drow = abs((pv - np.floor(pv)) - 0.5) if ok1 else 0.0
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
            else:
                unexplained.append({"x": x, "y": y, "u": u, "v": v, "bin_dist": float(d), "row_dist": float(drow), "ref": a[y, x].tolist(), "got": b[y, x].tolist()})
    out["classes"] = cls
    out["unexplained"] = len(unexplained)
    out["unexplained_examples"] = unexplained[:12]
    return out


