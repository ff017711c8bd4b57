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
  rlimit   the other way round: an r-limit is set and the twin rejects (writes background for) a ray the CPU path's formula keeps
  rlimit_row  the two formulas decide the pixel's rolling-shutter first pass differently (rlimit_first_pass_mask): another matrix row
  nan      the oracle's coordinate is NaN (refraction beyond total reflection): the CPU path samples at `NaN as i32` = 0, the twin writes background
  sentinel a coordinate (either pass) beyond +-99998: the twin marks invalid rays by the coordinate -99999 (.cl:403,535,616) and casts with C's `(int)`,
           so a diverged digital-lens iteration is "invalid" there and saturates in Rust
  feather  background mode 3, inside the feather zone: the twin scales the margin sample about (w-1, h-1), the CPU path about (w, h)
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


def run_reference_cl_host(name, pl, matrices, mesh=None):
    """The same kernel compiled for the host cores (oracle/_ref/gfw_ref_cl_<name>.host.so: oracle/build_ref_cl.py build_host +
    oracle/ref_cl_host.c): one call of the reference's undistort_image() per work-item of the output plane's NDRange.  No GPU."""
    path = os.path.join(ROOT, "oracle", "_ref", "gfw_ref_cl_%s.host.so" % name)
    if not os.path.exists(path):
        pytest.skip("host build of the reference OpenCL kernel not present (needs /root/reference at build time)")
    lib = C.CDLL(path)
    lib.gfw_ref_cl_run.argtypes = [C.c_void_p] * 6 + [C.c_int] * 3
    lib.gfw_ref_cl_run.restype = None
    src = np.ascontiguousarray(pl["src"])
    dst = pl["dst"].copy()
    prm = np.frombuffer(bytes(pl["params"]), dtype=np.uint8).copy()
    mat = np.ascontiguousarray(matrices, dtype=np.float32)
    drawing = np.zeros(16, dtype=np.uint8)
    mesh = np.zeros(16, dtype=np.float32) if mesh is None else np.ascontiguousarray(mesh, dtype=np.float32)
    ow, oh = pl["out_size"][0], pl["out_size"][1]
    lib.gfw_ref_cl_run(src.ctypes.data, dst.ctypes.data, prm.ctypes.data, mat.ctypes.data, drawing.ctypes.data, mesh.ctypes.data, ow, 0, oh)
    return dst


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


def in_feather_zone(p, u, v):
    """cpu_undistort.rs:582: the pixel blends a second, margin-scaled sample"""
    widthf, heightf = float(p.width) - 1.0, float(p.height) - 1.0
    feather = max(p.background_margin_feather * heightf, 0.0001)
    return (u > widthf - feather) or (u < feather) or (v > heightf - feather) or (v < feather)


def rlimit_first_pass_mask(fr, run_twin):
    """Pixels whose rolling-shutter FIRST pass (the projection with the middle row's matrix that picks the pixel's own row,
    cpu_undistort.rs:465-482) is decided differently by the two r-limit formulas: there the twins use different matrix rows for the
    second pass and the sample moves by a fraction of a pixel.  Found by running both on the same frame with the middle matrix alone
    (matrix_count = 1: the first pass is then the only pass) and comparing which pixels each leaves as background."""
    pl = dict(fr.planes[0])
    p = abi.KernelParams.from_buffer_copy(bytes(pl["params"]))
    mid = np.ascontiguousarray(fr.matrices[p.matrix_count // 2: p.matrix_count // 2 + 1], dtype=np.float32)
    p.matrix_count = 1
    pl["params"] = p
    dt = np.dtype(abi.PIXEL_TYPES[pl["pixel_type"]][1])
    w, h = pl["out_size"][0], pl["out_size"][1]
    dst = pl["dst"].copy()
    assert O.undistort_image(pl["src"], pl["size"], dst, pl["out_size"], p, pl["pixel_type"], fr.model, fr.digital, mid) == 1
    a = dst.view(dt).reshape(h, -1)[:, :w]
    b = run_twin(pl, mid).view(dt).reshape(h, -1)[:, :w]
    bg = dt.type(np.float32(p.background[0]) * np.float32(p.max_pixel_value))
    return (a == bg) != (b == bg)


def classify(fr, ref, got, interp, taus=(5e-5, 1e-4, 2e-4, 4e-4, 1e-3), tau_row=2e-3, rlimit_row_mask=None):
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
    cls = {"row": 0, "neg": 0, "invalid": 0, "nan": 0, "rlimit": 0, "rlimit_row": 0, "feather": 0, "sentinel": 0}
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
        p = pl["params"]
        bgpx = [float(np.float32(p.background[c]) * np.float32(p.max_pixel_value)) for c in range(n)]
        twin_wrote_bg = all(abs(float(bv) - g) < 1.0 for bv, g in zip(b[y, x].tolist(), bgpx))      # the twin's value is its background (truncated to the pixel type)
        if not ok:
            cls["invalid"] += 1                  # the oracle rejects the ray: w <= 0 agrees between the twins, the r-limit formula does not
            continue
        if u != u or v != v or (ok1 and p.matrix_count > 1 and (u1 != u1 or v1 != v1)):      # ... or the first pass's coordinate is NaN (its row pick then differs too)
            cls["nan"] += 1                      # NaN coordinate: the CPU path samples at the cast of NaN (0), the twin's `uv.x > -99998` test writes background (.cl:616)
            continue
        if max(abs(u), abs(v)) > 99998.0 or (ok1 and p.matrix_count > 1 and max(abs(u1), abs(v1)) > 99998.0):
            cls["sentinel"] += 1                 # a coordinate beyond the twin's "invalid" sentinel (-99999: .cl:535,616) or beyond i32 (its C cast is not Rust's saturating one)
            continue
        d = min(edge_dist(u, float(p.c[0])), edge_dist(v, float(p.c[1])))
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
            elif (u - off) < 0.0 or (v - off) < 0.0:
                cls["neg"] += 1
            elif p.r_limit > 0.0 and twin_wrote_bg:
                cls["rlimit"] += 1               # the twin's length((x, y) / w) > r_limit rejects a ray that x^2 + y^2 > r_limit^2 * w (sic) keeps
            elif rlimit_row_mask is not None and rlimit_row_mask[y, x]:
                cls["rlimit_row"] += 1           # the two r-limit formulas disagree on the pixel's first pass: a different matrix row
            elif p.background_mode == 3 and in_feather_zone(p, u, v):
                cls["feather"] += 1              # margin-with-feather: the twin scales the second sample about (w-1, h-1), the CPU path about (w, h) (.cl:622, cpu_undistort.rs:586)
            else:
                unexplained.append({"x": x, "y": y, "u": u, "v": v, "bin_dist": float(d), "row_dist": float(drow), "ref": a[y, x].tolist(), "got": b[y, x].tolist()})
    out["classes"] = cls
    out["unexplained"] = len(unexplained)
    out["unexplained_examples"] = unexplained[:12]
    return out


