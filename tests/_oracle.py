"""ctypes binding of the parity oracle (oracle/libgfw_oracle.so).  Test infrastructure only."""
import ctypes as C
import os
import subprocess

import numpy as np

from gyroflow_amd import abi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
ORACLE_SO = os.path.join(ORACLE_DIR, "libgfw_oracle.so")

_lib = None


def lib():
    global _lib
    if _lib is None:
        src = os.path.join(ORACLE_DIR, "gfw_oracle.c")
        if not os.path.exists(ORACLE_SO) or os.path.getmtime(ORACLE_SO) < os.path.getmtime(src):
            subprocess.check_call(["make", "-C", ORACLE_DIR, "-s"])
        L = C.CDLL(ORACLE_SO)
        vp = C.c_void_p
        L.gfw_oracle_undistort_image.argtypes = [C.POINTER(abi.Buffers), C.POINTER(abi.KernelParams), C.c_int, C.c_int,
                                                 C.c_int, vp, vp, C.c_size_t, C.c_int]
        L.gfw_oracle_undistort_image.restype = C.c_int
        L.gfw_oracle_undistort_coord.argtypes = [C.POINTER(abi.KernelParams), C.c_int, C.c_int, vp, vp, C.c_size_t,
                                                 C.c_float, C.c_float, vp]
        L.gfw_oracle_distort_point.argtypes = [C.c_int, C.POINTER(abi.KernelParams), C.c_float, C.c_float, C.c_float, vp]
        L.gfw_oracle_undistort_point.argtypes = [C.c_int, C.POINTER(abi.KernelParams), C.c_float, C.c_float, vp]
        L.gfw_oracle_undistort_point.restype = C.c_int
        L.gfw_oracle_stmap_undistort.argtypes = [C.POINTER(abi.KernelParams), C.c_int, C.c_int, vp, vp, C.c_size_t, C.c_int, C.c_int, vp, C.c_int]
        L.gfw_oracle_stmap_undistort.restype = C.c_int
        L.gfw_oracle_undistort_points.argtypes = [C.POINTER(abi.KernelParams), C.c_int, C.c_int, vp, C.c_size_t, C.c_int, vp, C.c_int, vp, C.c_int, vp, C.c_size_t, vp]
        L.gfw_oracle_undistort_points.restype = C.c_int
        L.gfw_oracle_undistort_frame.argtypes = [C.c_int, C.POINTER(abi.Buffers), C.POINTER(abi.KernelParams), C.POINTER(C.c_int), C.c_int, C.c_int, vp, C.c_int, C.c_int]
        L.gfw_oracle_undistort_frame.restype = C.c_int
        L.gfw_oracle_libm.argtypes = [C.c_int, vp, vp, C.c_size_t]
        L.gfw_oracle_num_threads.restype = C.c_int
        _lib = L
    return _lib


def host_buffers(src, in_size, dst, out_size, in_rect=None, out_rect=None, in_rot=None, out_rot=None):
    """Build a Buffers struct over two numpy uint8 arrays (BufferSource::Cpu)."""
    b = abi.Buffers()
    for d, arr, size, rect, rot in ((b.input, src, in_size, in_rect, in_rot), (b.output, dst, out_size, out_rect, out_rot)):
        d.width, d.height, d.stride = size
        d.has_rect = 1 if rect is not None else 0
        if rect is not None:
            for i in range(4):
                d.rect[i] = rect[i]
        d.has_rotation = 1 if rot is not None else 0
        d.rotation = rot or 0.0
        d.kind = abi.BUF_HOST
        d.data = arr.ctypes.data
        d.len = arr.nbytes
    return b


def undistort_image(src, in_size, dst, out_size, params, pixel_type, model, digital, matrices, mesh=None, nthreads=0):
    """Run the oracle's undistort_image_cpu restatement in place on ``dst``; returns its status."""
    b = host_buffers(src, in_size, dst, out_size)
    m = np.ascontiguousarray(matrices, dtype=np.float32)
    mesh_ptr, mesh_len = None, 0
    if mesh is not None and len(mesh):
        mesh = np.ascontiguousarray(mesh, dtype=np.float32)
        mesh_ptr, mesh_len = mesh.ctypes.data, mesh.size
    pid = abi.PIXEL_TYPES[pixel_type][0]
    return lib().gfw_oracle_undistort_image(C.byref(b), C.byref(params), pid, model, digital, m.ctypes.data,
                                            mesh_ptr, mesh_len, nthreads)


def undistort_coord(params, model, digital, matrices, x, y, mesh=None):
    out = np.zeros(3, dtype=np.float32)
    m = np.ascontiguousarray(matrices, dtype=np.float32)
    lib().gfw_oracle_undistort_coord(C.byref(params), model, digital, m.ctypes.data, None, 0, x, y, out.ctypes.data)
    return bool(out[0]), float(out[1]), float(out[2])


def run_frame(frame, nthreads=0):
    """Oracle over every plane of a SyntheticFrame; returns list of output arrays (copies)."""
    outs = []
    for pl in frame.planes:
        dst = pl["dst"].copy()
        st = undistort_image(pl["src"], pl["size"], dst, pl["out_size"], pl["params"], pl["pixel_type"],
                             frame.model, frame.digital, frame.matrices, nthreads=nthreads)
        assert st == 1, "oracle returned %d" % st
        outs.append(dst)
    return outs


class FrameRunner:
    """All planes of a frame through gfw_oracle_undistort_frame (ONE parallel region over the rows of every plane, static chunks): the form bench.py times as
    the CPU baseline.  Everything is marshalled once; ``run()`` is the timed call, ``outs`` the planes it wrote (uninitialised until the first run, so that
    the first touch of every output page happens on the thread that will keep writing it)."""

    def __init__(self, frame, nthreads=0, chunk=4):
        n = len(frame.planes)
        self.outs = [np.empty_like(pl["dst"]) for pl in frame.planes]
        self.fill = [pl["dst"] for pl in frame.planes]
        self.bufs = (abi.Buffers * n)(*[host_buffers(pl["src"], pl["size"], o, pl["out_size"]) for pl, o in zip(frame.planes, self.outs)])
        self.prm = (abi.KernelParams * n)(*[pl["params"] for pl in frame.planes])
        self.typ = (C.c_int * n)(*[abi.PIXEL_TYPES[pl["pixel_type"]][0] for pl in frame.planes])
        self.m = np.ascontiguousarray(frame.matrices, dtype=np.float32)
        self.args = (n, self.bufs, self.prm, self.typ, frame.model, frame.digital, self.m.ctypes.data, nthreads, chunk)
        self.fn = lib().gfw_oracle_undistort_frame

    def run(self):
        st = self.fn(*self.args)
        assert st == 1, "oracle returned %d" % st
        return self.outs


def run_frame_fast(frame, nthreads=0):
    """run_frame through the whole-frame entry point (bytes the warp never writes are taken from the frame's ``dst`` like run_frame's copies)."""
    r = FrameRunner(frame, nthreads)
    for o, f in zip(r.outs, r.fill):
        o[...] = f
    return r.run()


def stmap_undistort(params, model, digital, matrices, width, height, nthreads=0):
    """Oracle restatement of the stmap.rs 'undist' closure; returns float32 [height][width][2] (0 where None)."""
    m = np.ascontiguousarray(matrices, dtype=np.float32)
    coords = np.zeros((height, width, 2), dtype=np.float32)
    lib().gfw_oracle_stmap_undistort(C.byref(params), model, digital, m.ctypes.data, None, 0, width, height, coords.ctypes.data, nthreads)
    return coords


def undistort_points(params, model, digital, rotations, points=None, grid=None, shifts=None, index_mode=0, mesh=None):
    """Oracle restatement of `undistort_points` (cpu_undistort.rs:652-858, lens_correction_amount == 1)."""
    rot = np.ascontiguousarray(rotations, dtype=np.float32).reshape(-1, 9)
    if points is not None:
        pts = np.ascontiguousarray(points, dtype=np.float32).reshape(-1, 2)
        n, gw, pp, shape = pts.shape[0], 0, pts.ctypes.data, (pts.shape[0], 2)
    else:
        gw, gh = grid
        n, pp, shape = gw * gh, None, (gh, gw, 2)
    out = np.zeros(shape, dtype=np.float32)
    sp = None
    if shifts is not None:
        shifts = np.ascontiguousarray(shifts, dtype=np.float32).reshape(-1, 5)
        sp = shifts.ctypes.data
    meshp, meshn = None, 0
    if mesh is not None and len(mesh):
        mesh = np.ascontiguousarray(mesh, dtype=np.float64)
        meshp, meshn = mesh.ctypes.data, mesh.size
    lib().gfw_oracle_undistort_points(C.byref(params), model, digital, pp, n, gw, rot.ctypes.data, rot.shape[0], sp, index_mode,
                                      meshp, meshn, out.ctypes.data)
    return out


def undistort_point(model, params, x, y):
    """One lens-model inverse (distortion_models/*.rs `undistort_point`): (ok, x, y)."""
    out = np.zeros(2, dtype=np.float32)
    ok = lib().gfw_oracle_undistort_point(model, C.byref(params), float(x), float(y), out.ctypes.data)
    return bool(ok), float(out[0]), float(out[1])
