"""Thin ctypes driver over the C ABI (include/gfwarp.h).

``Backend`` plays the role of the reference's ``OclWrapper`` / ``WgpuWrapper`` objects
(src/core/gpu/opencl.rs:178,330): construct once per (size, pixel type, lens model) key, then call
``undistort_image`` per plane or ``undistort_frame`` per frame.  All pixel work happens in
libgfwarp.so on the GPU; this module only marshals pointers.
"""
import ctypes as C

import numpy as np

from . import abi


class GfwError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("%s (%d): %s" % (abi.ERRORS.get(code, "?"), code, msg))
        self.code = code
        self.name = abi.ERRORS.get(code, "?")


def _desc(d, size, kind, ptr, nbytes, rect=None, rotation=None):
    d.width, d.height, d.stride = size
    d.has_rect = 1 if rect is not None else 0
    if rect is not None:
        for i in range(4):
            d.rect[i] = rect[i]
    d.has_rotation = 1 if rotation is not None else 0
    d.rotation = rotation or 0.0
    d.kind = kind
    d.data = ptr
    d.len = nbytes


def host_buffers(src, in_size, dst, out_size, in_rect=None, out_rect=None, in_rot=None, out_rot=None):
    """``Buffers`` over two numpy uint8 arrays (BufferSource::Cpu)."""
    b = abi.Buffers()
    _desc(b.input, in_size, abi.BUF_HOST, src.ctypes.data, src.nbytes, in_rect, in_rot)
    _desc(b.output, out_size, abi.BUF_HOST, dst.ctypes.data, dst.nbytes, out_rect, out_rot)
    return b


def device_buffers(src_ptr, src_len, in_size, dst_ptr, dst_len, out_size, in_rect=None, out_rect=None):
    """``Buffers`` over raw HIP device pointers (the CUDABuffer analogue)."""
    b = abi.Buffers()
    _desc(b.input, in_size, abi.BUF_HIP_DEVICE, src_ptr, src_len, in_rect)
    _desc(b.output, out_size, abi.BUF_HIP_DEVICE, dst_ptr, dst_len, out_rect)
    return b


_last_backend = ""


def last_backend():
    return _last_backend


class Backend:
    def __init__(self, params, pixel_type, model, digital, buffers, drawing_len=0):
        self.lib = abi.load_library()
        pid = abi.PIXEL_TYPES[pixel_type][0] if isinstance(pixel_type, str) else pixel_type
        self.ctx = self.lib.gfw_create(C.byref(params), pid, model, digital, C.byref(buffers), drawing_len)
        if not self.ctx:
            raise GfwError(-100, self.lib.gfw_last_error().decode())

    def close(self):
        if getattr(self, "ctx", None):
            self.lib.gfw_destroy(self.ctx)
            self.ctx = None

    __del__ = close

    def _check(self, rc):
        global _last_backend
        if rc != 0:
            raise GfwError(rc, self.lib.gfw_last_error().decode())
        _last_backend = self.lib.gfw_last_backend(self.ctx).decode()

    @staticmethod
    def last_backend_of(be):
        """gfw_last_backend of this context (the module-level last_backend() names whichever context was called last)."""
        return be.lib.gfw_last_backend(be.ctx).decode()

    def set_option(self, opt, value):
        self._check(self.lib.gfw_set_option(self.ctx, opt, value))

    def set_stream(self, stream_ptr):
        self._check(self.lib.gfw_set_stream(self.ctx, stream_ptr))

    def get_stream(self):
        """hipStream_t the context enqueues on (an int), after whatever it holds for a frame or a launch has been enqueued (gfw_get_stream flushes)."""
        return self.lib.gfw_get_stream(self.ctx)

    def set_frame_checksums(self, d_sums_ptr, count):
        """From now on frame k submitted on the context adds the checksum of the bytes it writes to the device word d_sums_ptr[k % count] (gfw_set_frame_checksums);
        (0, 0) turns it off."""
        self._check(self.lib.gfw_set_frame_checksums(self.ctx, d_sums_ptr, count))

    def jit_status(self):
        """(state, compile milliseconds, compiler log) of the context's run-time specialised kernel; state 0 none / unavailable,
        1 compiling, 2 ready, 3 failed (gfw_jit_status)."""
        ms, log = C.c_double(0.0), C.create_string_buffer(4096)
        st = self.lib.gfw_jit_status(self.ctx, C.byref(ms), log, 4096)
        return st, ms.value, log.value.decode(errors="replace")

    def get_profile_frames(self, reset=True):
        """(kernel milliseconds, launches, frames those launches covered) since the last reset (needs OPT_PROFILE)."""
        ms, n, f = C.c_double(0.0), C.c_int64(0), C.c_int64(0)
        self._check(self.lib.gfw_get_profile_frames(self.ctx, C.byref(ms), C.byref(n), C.byref(f), 1 if reset else 0))
        return ms.value, n.value, f.value

    def get_profile(self, reset=True):
        """(kernel milliseconds, launches) accumulated since the last reset (needs OPT_PROFILE)."""
        ms, n = C.c_double(0.0), C.c_int64(0)
        self._check(self.lib.gfw_get_profile(self.ctx, C.byref(ms), C.byref(n), 1 if reset else 0))
        return ms.value, n.value

    def get_audit(self, reset=False):
        """First-pass audit: (certified, certified_but_wrong, queued, queue_overflow, max |approx - exact| in pixels)."""
        arr = (C.c_ulonglong * 8)()
        self._check(self.lib.gfw_get_audit(self.ctx, C.byref(arr), 1 if reset else 0))
        gap = float(np.array([int(arr[4]) & 0xFFFFFFFF], dtype=np.uint32).view(np.float32)[0])
        return int(arr[0]), int(arr[1]), int(arr[2]), int(arr[3]), gap

    def get_audit_full(self, reset=False):
        """All audit words by name (first pass + address audit)."""
        arr = (C.c_ulonglong * 8)()
        self._check(self.lib.gfw_get_audit(self.ctx, C.byref(arr), 1 if reset else 0))
        gap = float(np.array([int(arr[4]) & 0xFFFFFFFF], dtype=np.uint32).view(np.float32)[0])
        return {"certified1": int(arr[0]), "certified1_wrong": int(arr[1]), "queued1": int(arr[2]), "queue_overflow": int(arr[3]),
                "pass1_gap_px": gap, "out_of_range": int(arr[5]),
                "pass1_eps_px": float(np.array([int(arr[6]) & 0xFFFFFFFF], dtype=np.uint32).view(np.float32)[0])}

    def set_quaternion_tracks(self, org, smoothed):
        """Upload (timestamps_us int64, quaternions f64[n,4]) tracks once per clip (device matrix builder)."""
        ot, oq = np.ascontiguousarray(org[0], dtype=np.int64), np.ascontiguousarray(org[1], dtype=np.float64)
        st, sq = np.ascontiguousarray(smoothed[0], dtype=np.int64), np.ascontiguousarray(smoothed[1], dtype=np.float64)
        self._check(self.lib.gfw_set_quaternion_tracks(self.ctx, ot.ctypes.data, oq.ctypes.data, len(ot), st.ctypes.data, sq.ctypes.data, len(st)))

    def set_sync_offsets(self, duration_ms, timestamps_us=(), offsets_ms=()):
        """The clip's gyro/video sync offsets (GyroSource.offsets_adjusted) and duration (gyro_source/mod.rs:857-860)."""
        ts = np.ascontiguousarray(timestamps_us, dtype=np.int64)
        ov = np.ascontiguousarray(offsets_ms, dtype=np.float64)
        assert ts.shape == ov.shape
        self._check(self.lib.gfw_set_sync_offsets(self.ctx, float(duration_ms), ts.ctypes.data if len(ts) else None, ov.ctypes.data if len(ts) else None, len(ts)))

    def build_matrices(self, nk, timestamp_ms, frame_readout_time_ms, rows, readout_dim, video_rotation_deg=0.0,
                       framebuffer_inverted=False, per_frame_offset_ms=0.0, out_ptr=None, suppress_rotation=0, stab=None):
        """Build one frame's packed rows on the device; returns the device pointer (context-owned unless out_ptr given).
        stab: None or dict(offset, sensor_size, crop_area, pixel_pitch, width, height, ibis=[n][4], ois=[n][4])."""
        t = abi.FrameTiming()
        t.timestamp_ms, t.per_frame_time_offset_ms, t.frame_readout_time_ms = timestamp_ms, per_frame_offset_ms, frame_readout_time_ms
        for i, v in enumerate(np.asarray(nk, dtype=np.float64).reshape(9)):
            t.new_k[i] = v
        t.video_rotation_deg, t.rows, t.readout_dim = video_rotation_deg, rows, readout_dim
        t.framebuffer_inverted = 1 if framebuffer_inverted else 0
        t.suppress_rotation = int(suppress_rotation)
        ptr = C.c_void_p(0)
        if stab is None:
            self._check(self.lib.gfw_build_matrices(self.ctx, C.byref(t), out_ptr, C.byref(ptr)))
            return ptr.value
        st = abi.FrameStab()
        st.offset = stab["offset"]
        st.sensor_size[0], st.sensor_size[1] = stab["sensor_size"]
        for i in range(4):
            st.crop_area[i] = stab["crop_area"][i]
        st.pixel_pitch[0], st.pixel_pitch[1] = stab["pixel_pitch"]
        st.width, st.height = stab["width"], stab["height"]
        ibis = np.ascontiguousarray(stab["ibis"], dtype=np.float64).reshape(-1, 4)
        ois = np.ascontiguousarray(stab["ois"], dtype=np.float64).reshape(-1, 4)
        st.ibis_count, st.ois_count = ibis.shape[0], ois.shape[0]
        st.ibis, st.ois = ibis.ctypes.data, ois.ctypes.data
        self._check(self.lib.gfw_build_matrices_stab(self.ctx, C.byref(t), C.byref(st), out_ptr, C.byref(ptr)))
        return ptr.value

    def build_matrices_batch(self, nk, timestamps_ms, frame_readout_time_ms, rows, readout_dim, video_rotation_deg=0.0,
                             framebuffer_inverted=False, per_frame_offset_ms=0.0):
        """Tables of several upcoming frames in one launch (gfw_build_matrices_batch); returns the device pointers."""
        n = len(timestamps_ms)
        arr = (abi.FrameTiming * n)()
        nkf = np.asarray(nk, dtype=np.float64).reshape(9)
        for k, ts in enumerate(timestamps_ms):
            t = arr[k]
            t.timestamp_ms, t.per_frame_time_offset_ms, t.frame_readout_time_ms = ts, per_frame_offset_ms, frame_readout_time_ms
            for i in range(9):
                t.new_k[i] = nkf[i]
            t.video_rotation_deg, t.rows, t.readout_dim = video_rotation_deg, rows, readout_dim
            t.framebuffer_inverted = 1 if framebuffer_inverted else 0
        ptrs = (C.c_void_p * n)()
        self._check(self.lib.gfw_build_matrices_batch(self.ctx, arr, n, ptrs))
        return [p for p in ptrs]

    def stmap_undistort(self, params, matrices, width, height, mesh=None):
        """STMap 'undist' coordinates (stmap.rs:87-109) as a float32 array [height][width][2] (0 where None)."""
        m = np.ascontiguousarray(matrices, dtype=np.float32)
        coords = np.zeros((height, width, 2), dtype=np.float32)
        meshp, meshn = None, 0
        if mesh is not None and len(mesh):
            mesh = np.ascontiguousarray(mesh, dtype=np.float32)
            meshp, meshn = mesh.ctypes.data, mesh.size
        self._check(self.lib.gfw_stmap_undistort(self.ctx, C.byref(params), m.ctypes.data, m.shape[0], meshp, meshn, width, height, coords.ctypes.data, 0))
        return coords

    def undistort_points(self, params, rotations, points=None, grid=None, shifts=None, index_mode=0, mesh=None):
        """Inverse point map (cpu_undistort.rs:652-858, lens_correction_amount == 1).

        ``points``: [n][2] float32, or None with ``grid=(w, h)`` for the pixel grid.  ``rotations``: [count][9] float32
        (`new_k * R` row-major), ``shifts``: None or [count][5], ``mesh``: None or float64 mesh data.
        Returns float32 [n][2] (or [h][w][2] for a grid)."""
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
            assert shifts.shape[0] == rot.shape[0]
            sp = shifts.ctypes.data
        meshp, meshn = None, 0
        if mesh is not None and len(mesh):
            mesh = np.ascontiguousarray(mesh, dtype=np.float64)
            meshp, meshn = mesh.ctypes.data, mesh.size
        self._check(self.lib.gfw_undistort_points(self.ctx, C.byref(params), pp, n, gw, rot.ctypes.data, rot.shape[0], sp,
                                                  index_mode, meshp, meshn, out.ctypes.data, 0))
        return out

    def synchronize(self):
        self._check(self.lib.gfw_synchronize(self.ctx))

    def flush(self):
        """Enqueue whatever plane coalescing holds for a frame this context belongs to (gfw_flush); does not wait for the GPU."""
        self._check(self.lib.gfw_flush(self.ctx))

    def undistort_image(self, buffers, params, matrices, mesh=None, matrix_count=None):
        """One plane (OclWrapper::undistort_image).  ``matrices``: numpy [rows][14] f32, or a device pointer int."""
        if isinstance(matrices, np.ndarray):
            m = np.ascontiguousarray(matrices, dtype=np.float32)
            mp, mc = m.ctypes.data, m.shape[0]
        else:
            mp, mc = matrices, matrix_count
        meshp, meshn = None, 0
        if mesh is not None and len(mesh):
            mesh = np.ascontiguousarray(mesh, dtype=np.float32)
            meshp, meshn = mesh.ctypes.data, mesh.size
        self._check(self.lib.gfw_undistort_image(self.ctx, C.byref(buffers), C.byref(params), mp, mc, None, 0, meshp, meshn))

    def undistort_frame(self, planes, params, pixel_types, matrices, mesh=None, matrix_count=None):
        """All planes of a frame in one call (additive entry point)."""
        n = len(planes)
        barr = (abi.Buffers * n)(*planes)
        parr = (abi.KernelParams * n)(*params)
        tarr = (C.c_int * n)(*[abi.PIXEL_TYPES[t][0] if isinstance(t, str) else t for t in pixel_types])
        if isinstance(matrices, np.ndarray):
            m = np.ascontiguousarray(matrices, dtype=np.float32)
            mp, mc = m.ctypes.data, m.shape[0]
        else:
            mp, mc = matrices, matrix_count
        meshp, meshn = None, 0
        if mesh is not None and len(mesh):
            mesh = np.ascontiguousarray(mesh, dtype=np.float32)
            meshp, meshn = mesh.ctypes.data, mesh.size
        self._check(self.lib.gfw_undistort_frame(self.ctx, n, barr, parr, tarr, mp, mc, meshp, meshn))


def pack_matrices(matrices):
    """[rows][14] f32 -> libgfwarp's packed [rows][16] layout (host libm trig for the IBIS slots)."""
    m = np.ascontiguousarray(matrices, dtype=np.float32)
    out = np.empty((m.shape[0], 16), dtype=np.float32)
    rc = abi.load_library().gfw_pack_matrices(m.ctypes.data, m.shape[0], out.ctypes.data)
    if rc != 0:
        raise GfwError(rc, abi.load_library().gfw_last_error().decode())
    return out


class FrameCall:
    """Pre-marshalled ``gfw_undistort_frame`` call (the per-frame hot loop of a renderer keeps these around so that
    no ctypes objects are built per frame)."""

    def __init__(self, backend, planes, params, pixel_types, matrices, matrix_count=None):
        n = len(planes)
        self.be, self.n = backend, n
        self.barr = (abi.Buffers * n)(*planes)
        self.parr = (abi.KernelParams * n)(*params)
        self.tarr = (C.c_int * n)(*[abi.PIXEL_TYPES[t][0] if isinstance(t, str) else t for t in pixel_types])
        if isinstance(matrices, np.ndarray):
            self.m = np.ascontiguousarray(matrices, dtype=np.float32)
            self.mp, self.mc = self.m.ctypes.data, self.m.shape[0]
        else:                                       # device pointer (GFW_OPT_MATRICES_ON_DEVICE)
            self.mp, self.mc = matrices, matrix_count
        self.fn = backend.lib.gfw_undistort_frame

    def __call__(self):
        rc = self.fn(self.be.ctx, self.n, self.barr, self.parr, self.tarr, self.mp, self.mc, None, 0)
        if rc != 0:
            self.be._check(rc)


class PlaneCalls:
    """Pre-marshalled per-plane ``gfw_undistort_image`` calls of ONE frame, plane p through backend p — the call sequence of the reference's render loop
    (one process_pixels per plane, each plane its own Stabilization / backend object, src/rendering/mod.rs:494-545)."""

    def __init__(self, backends, planes, params, matrices, matrix_count=None):
        if isinstance(matrices, np.ndarray):
            self.m = np.ascontiguousarray(matrices, dtype=np.float32)
            mp, mc = self.m.ctypes.data, self.m.shape[0]
        else:
            mp, mc = matrices, matrix_count
        self.keep = [(abi.Buffers.from_buffer_copy(b), abi.KernelParams.from_buffer_copy(p)) for b, p in zip(planes, params)]
        for i, (_, p) in enumerate(self.keep):
            p.plane_index = i
        self.items = [(be.lib.gfw_undistort_image, be.ctx, C.byref(b), C.byref(p), mp, mc, be) for be, (b, p) in zip(backends, self.keep)]

    def __call__(self):
        for fn, ctx, b, p, mp, mc, be in self.items:
            rc = fn(ctx, b, p, mp, mc, None, 0, None, 0)
            if rc != 0:
                be._check(rc)


class ClipCall:
    """Pre-marshalled ``gfw_undistort_clip`` call: ``frames`` is a list of per-frame plane lists (``Buffers``), ``matrices`` a list
    of per-frame tables (device pointers, or numpy [rows][14] arrays), ``params`` / ``pixel_types`` are shared by all frames."""

    def __init__(self, backend, frames, params, pixel_types, matrices, matrix_count=None):
        nf, n = len(frames), len(frames[0])
        self.be, self.nf, self.n = backend, nf, n
        self.barr = (abi.Buffers * (nf * n))(*[b for fr in frames for b in fr])
        self.parr = (abi.KernelParams * n)(*params)
        self.tarr = (C.c_int * n)(*[abi.PIXEL_TYPES[t][0] if isinstance(t, str) else t for t in pixel_types])
        self.keep, ptrs = [], []
        for m in matrices:
            if isinstance(m, np.ndarray):
                m = np.ascontiguousarray(m, dtype=np.float32)
                self.keep.append(m)
                ptrs.append(m.ctypes.data)
                matrix_count = m.shape[0]
            else:
                ptrs.append(m)
        self.marr = (C.c_void_p * nf)(*ptrs)
        self.mc = matrix_count
        self.fn = backend.lib.gfw_undistort_clip

    def __call__(self):
        rc = self.fn(self.be.ctx, self.nf, self.n, self.barr, self.parr, self.tarr, self.marr, self.mc)
        if rc != 0:
            self.be._check(rc)


def run_plane(src, in_size, dst, out_size, params, pixel_type, model, digital, matrices, mesh=None, **rects):
    """Convenience: create a backend, warp one HOST plane in place into ``dst``."""
    b = host_buffers(src, in_size, dst, out_size, **rects)
    be = Backend(params, pixel_type, model, digital, b)
    try:
        be.undistort_image(b, params, matrices, mesh)
    finally:
        be.close()


def run_frame(frame, fused=True, per_plane=False, variant=None, jit=None):
    """Warp every plane of a ``synthetic.SyntheticFrame`` from HOST buffers; returns output copies.

    fused=False forces the generic per-plane kernel (GFW_OPT_KERNEL_VARIANT = 1); per_plane=True issues one
    ``gfw_undistort_image`` per plane, the way the reference's render loop does; jit sets GFW_OPT_JIT (2: the frame waits
    for its run-time specialised kernel)."""
    outs = [pl["dst"].copy() for pl in frame.planes]
    bufs = [host_buffers(pl["src"], pl["size"], o, pl["out_size"]) for pl, o in zip(frame.planes, outs)]
    params = [pl["params"] for pl in frame.planes]
    types = [pl["pixel_type"] for pl in frame.planes]
    if per_plane:
        for b, p, t in zip(bufs, params, types):
            be = Backend(p, t, frame.model, frame.digital, b)
            try:
                if not fused:
                    be.set_option(abi.OPT_KERNEL_VARIANT, 1)
                be.undistort_image(b, p, frame.matrices)
            finally:
                be.close()
        return outs
    be = Backend(params[0], types[0], frame.model, frame.digital, bufs[0])
    try:
        if not fused:
            be.set_option(abi.OPT_KERNEL_VARIANT, 1)
        elif variant is not None:
            be.set_option(abi.OPT_KERNEL_VARIANT, variant)
        if jit is not None:
            be.set_option(abi.OPT_JIT, jit)
        be.undistort_frame(bufs, params, types, frame.matrices)
    finally:
        be.close()
    return outs
