"""Host-side mirror of the reference's operator surface for the warp path.

Names, argument meaning and error behaviour follow ``src/core/stabilization/mod.rs`` so that tests read like uses of
the reference:

    stab = Stabilization()
    stab.interpolation = Interpolation.Bilinear
    stab.init_size((w, h), (ow, oh))                       # mod.rs:375
    stab.set_compute_params(ComputeParams(...))            # mod.rs:204
    transform = stab.get_frame_transform_at("Luma16", timestamp_us, None, buffers)      # mod.rs:253
    info = stab.process_pixels("Luma16", timestamp_us, None, buffers, transform)        # mod.rs:612

``process_pixels`` validates exactly like the reference (SizeTooSmall / SizeMismatch / InvalidStride /
NoStabilizationData), then dispatches to the HIP backend object (``warp.Backend`` = the OclWrapper/WgpuWrapper slot,
mod.rs:643-702).  There is no CPU arm: a missing GPU surfaces as an error, never as a silent fallback.

``FrameTransform.at_timestamp`` is the host (float64) statement of frame_transform.rs:165-350 for the synthetic
clips used here (quaternion track -> per-row ``inv(new_k * R)``); the device version of the same step is
``gfw_build_matrices`` (next row of SURVEY.md section 8f).
"""
import collections
import enum

import numpy as np

from . import abi, warp
from . import synthetic as S


class Interpolation(enum.IntEnum):          # mod.rs:25-34
    Bilinear = 2
    Bicubic = 4
    Lanczos4 = 8
    RobidouxSharp = 10
    Robidoux = 11
    Mitchell = 12
    CatmullRom = 13


class GyroflowCoreError(Exception):
    """``GyroflowCoreError`` (src/core/lib.rs:2099-2141); ``kind`` is the variant name."""

    def __init__(self, kind, detail=""):
        super().__init__("%s%s" % (kind, (": " + str(detail)) if detail != "" else ""))
        self.kind = kind
        self.detail = detail


_CODE_TO_KIND = {-1: "SizeTooSmall", -2: "SizeMismatch", -3: "InvalidStride", -4: "NoStabilizationData",
                 -5: "InputBufferEmpty", -6: "OutputBufferEmpty"}


class BufferDescription:
    """``BufferDescription`` (gpu/mod.rs:17-24).  ``data`` is a 1-D numpy uint8 array (BufferSource::Cpu) or a
    ``(device_ptr, nbytes)`` tuple (the CUDABuffer analogue)."""

    def __init__(self, size, data, rect=None, rotation=None):
        self.size, self.data, self.rect, self.rotation = tuple(size), data, rect, rotation

    def _fill(self, d):
        if isinstance(self.data, np.ndarray):
            warp._desc(d, self.size, abi.BUF_HOST, self.data.ctypes.data, self.data.nbytes, self.rect, self.rotation)
        elif self.data is None:
            warp._desc(d, self.size, abi.BUF_NONE, None, 0, self.rect, self.rotation)
        else:
            warp._desc(d, self.size, abi.BUF_HIP_DEVICE, self.data[0], self.data[1], self.rect, self.rotation)

    def nbytes(self):
        if isinstance(self.data, np.ndarray):
            return self.data.nbytes
        return 0 if self.data is None else self.data[1]


class Buffers:
    """``Buffers`` (gpu/mod.rs:25-28)."""

    def __init__(self, input, output):
        self.input, self.output = input, output

    def to_abi(self):
        b = abi.Buffers()
        self.input._fill(b.input)
        self.output._fill(b.output)
        return b

    def get_checksum(self):
        return hash((self.input.size, self.input.rect, self.output.size, self.output.rect,
                     isinstance(self.input.data, np.ndarray), isinstance(self.output.data, np.ndarray)))


class ComputeParams:
    """The slice of ``ComputeParams`` (compute_params.rs:71-138) the warp path consumes."""

    def __init__(self, lens, distortion_model="opencv_fisheye", digital_lens=None, digital_lens_params=(),
                 width=0, height=0, output_width=0, output_height=0, fov_scale=1.0, fovs=(), frame_readout_time=0.0,
                 horizontal_rs=False, background=(0.0, 0.0, 0.0, 0.0), background_mode=0, background_margin=0.0,
                 background_margin_feather=0.0, lens_correction_amount=1.0, light_refraction_coefficient=1.0,
                 adaptive_zoom_center_offset=(0.0, 0.0), scaled_fps=30.0, org_quat_at=None, smoothed_quat_at=None,
                 video_rotation=0.0, framebuffer_inverted=False):
        self.lens = lens
        self.distortion_model, self.digital_lens, self.digital_lens_params = distortion_model, digital_lens, digital_lens_params
        self.width, self.height, self.output_width, self.output_height = width, height, output_width, output_height
        self.fov_scale, self.fovs = fov_scale, list(fovs)
        self.frame_readout_time, self.horizontal_rs = frame_readout_time, horizontal_rs
        self.background, self.background_mode = tuple(background), background_mode
        self.background_margin, self.background_margin_feather = background_margin, background_margin_feather
        self.lens_correction_amount = lens_correction_amount
        self.light_refraction_coefficient = light_refraction_coefficient
        self.adaptive_zoom_center_offset = adaptive_zoom_center_offset
        self.scaled_fps = scaled_fps
        ident = lambda t: np.array([1.0, 0.0, 0.0, 0.0])
        self.org_quat_at = org_quat_at or ident
        self.smoothed_quat_at = smoothed_quat_at or ident
        self.video_rotation, self.framebuffer_inverted = video_rotation, framebuffer_inverted


class FrameTransform:
    """``FrameTransform`` (frame_transform.rs:12-19)."""

    def __init__(self, matrices, kernel_params, fov=1.0, minimal_fov=1.0, focal_length=None, mesh_data=()):
        self.matrices, self.kernel_params = matrices, kernel_params
        self.fov, self.minimal_fov, self.focal_length, self.mesh_data = fov, minimal_fov, focal_length, list(mesh_data)

    @staticmethod
    def get_fov(params, frame):
        """frame_transform.rs:52-58"""
        fovs = params.fovs
        base = (fovs[frame] if frame < len(fovs) else (fovs[-1] if len(fovs) > 1 else 1.0)) * params.fov_scale
        fov = max(base, 0.001)
        return fov * params.width / max(params.output_width, 1)

    @staticmethod
    def at_timestamp(params, timestamp_ms, frame):
        """frame_transform.rs:165-350 for clips without IBIS/mesh/keyframes: float64 on the host."""
        fov = FrameTransform.get_fov(params, frame)
        lens = params.lens
        nk = S.new_k(lens, fov, params.output_width, params.output_height)                  # :37-51
        frt = params.frame_readout_time
        rows = (params.width if params.horizontal_rs else params.height) if abs(frt) > 0.0 else 1   # :247
        row_t = frt / (params.width if params.horizontal_rs else params.height)
        start_ts = timestamp_ms - frt / 2.0
        quat1 = params.org_quat_at(timestamp_ms)
        quat1 = np.array([quat1[0], -quat1[1], -quat1[2], -quat1[3]]) / np.dot(quat1, quat1)     # inverse
        smoothed = params.smoothed_quat_at(timestamp_ms)
        a = np.radians(params.video_rotation)
        image_rotation = np.array([[np.cos(a), -np.sin(a), 0.0], [np.sin(a), np.cos(a), 0.0], [0.0, 0.0, 1.0]])
        out = np.zeros((rows, 14), dtype=np.float32)
        for y in range(rows):
            qt = start_ts + row_t * y if abs(frt) > 0.0 else start_ts
            q = S.quat_mul(smoothed, S.quat_mul(quat1, params.org_quat_at(qt)))
            r = image_rotation @ S.quat_to_matrix(q)
            if params.framebuffer_inverted:                                                  # :261-267
                r[0, 2] *= -1.0; r[1, 2] *= -1.0; r[2, 0] *= -1.0; r[2, 1] *= -1.0
            else:
                r[0, 1] *= -1.0; r[0, 2] *= -1.0; r[1, 0] *= -1.0; r[2, 0] *= -1.0
            out[y, :9] = np.linalg.pinv(nk @ r, rcond=1e-6).reshape(9).astype(np.float32)   # :296
        czx, czy = params.adaptive_zoom_center_offset
        if params.framebuffer_inverted:
            czy *= -1.0
        kp = S.base_kernel_params(
            lens, fov, rows, lens_correction_amount=params.lens_correction_amount,
            background_mode=params.background_mode, background_margin=params.background_margin,
            background_margin_feather=params.background_margin_feather,
            translation2d=(czx * params.width / fov, czy * params.height / fov),
            digital_lens_params=list(params.digital_lens_params),
            light_refraction_coefficient=params.light_refraction_coefficient)
        return FrameTransform(out, kp, fov=fov)


ProcessedInfo = collections.namedtuple("ProcessedInfo", "fov minimal_fov focal_length backend")


class Stabilization:
    """``Stabilization`` (mod.rs:169-192): one instance per plane in the render loop (rendering/mod.rs:494)."""

    def __init__(self):
        self.stab_data = {}
        self.size = (0, 0)
        self.output_size = (0, 0)
        self.interpolation = Interpolation.Bilinear
        self.kernel_flags = 0
        self.compute_params = None
        self.cache_frame_transform = False
        self.initialized_backend = None           # ("HIP", key) once ensure_ready_for_processing ran
        self._backends = collections.OrderedDict()   # LRU(15) of backend objects, mod.rs:59-66

    # -- mod.rs:204 / :375 ------------------------------------------------------------------------
    def set_compute_params(self, params):
        self.stab_data.clear()
        self.compute_params = params

    def init_size(self, size, output_size):
        self.initialized_backend = None
        self.size, self.output_size = tuple(size), tuple(output_size)
        self.stab_data.clear()

    @staticmethod
    def get_rect(desc):                                                  # mod.rs:209-224
        return tuple(desc.rect) if desc.rect is not None else (0, 0, desc.size[0], desc.size[1])

    def get_kernel_flags(self, frame, buffers):                          # mod.rs:226-251
        f = self.kernel_flags
        cp = self.compute_params

        def setf(bit, cond):
            return (f | bit) if cond else (f & ~bit)
        f = setf(abi.FLAG_HAS_DIGITAL_LENS, cp.digital_lens is not None)
        f = setf(abi.FLAG_HORIZONTAL_RS, cp.horizontal_rs)
        f = setf(abi.FLAG_HAS_SOURCE_RECT, buffers.input.rect is not None or self.size != tuple(buffers.input.size[:2]))
        f = setf(abi.FLAG_HAS_OUTPUT_RECT, buffers.output.rect is not None or self.output_size != tuple(buffers.output.size[:2]))
        f = setf(abi.FLAG_FRAMEBUFFER_INVERTED, cp.framebuffer_inverted)
        f = setf(abi.FLAG_ANY_UNDERWATER, cp.light_refraction_coefficient != 1.0 and cp.light_refraction_coefficient > 0.0)
        return f

    def get_frame_transform_at(self, pixel_type, timestamp_us, frame, buffers):       # mod.rs:253-326
        ts_ms = timestamp_us / 1000.0
        if frame is None:
            frame = int(round(ts_ms * self.compute_params.scaled_fps / 1000.0))
        t = FrameTransform.at_timestamp(self.compute_params, ts_ms, frame)
        t.kernel_params = S.plane_kernel_params(
            t.kernel_params, pixel_type, self.size, self.output_size,
            (buffers.input.size[0], buffers.input.size[1], buffers.input.size[2], buffers.input.rect, buffers.input.rotation),
            (buffers.output.size[0], buffers.output.size[1], buffers.output.size[2], buffers.output.rect, buffers.output.rotation),
            interpolation=int(self.interpolation), flags=self.get_kernel_flags(frame, buffers), background=self.compute_params.background)
        return t

    def get_current_key(self, buffers):                                   # mod.rs:355-373
        cp = self.compute_params
        flags = self.get_kernel_flags(0, buffers) & ~abi.FLAG_FILL_WITH_BACKGROUND
        return (buffers.get_checksum(), cp.distortion_model, cp.digital_lens, int(self.interpolation), flags, self.size, self.output_size)

    def ensure_ready_for_processing(self, pixel_type, timestamp_us, frame, buffers):   # mod.rs:567-611
        key = (pixel_type,) + self.get_current_key(buffers)
        if key not in self._backends:
            t = self.get_frame_transform_at(pixel_type, timestamp_us, frame, buffers)
            cp = self.compute_params
            be = warp.Backend(t.kernel_params, pixel_type, abi.MODELS[cp.distortion_model],
                              abi.MODELS[cp.digital_lens] if cp.digital_lens else 0, buffers.to_abi())
            self._backends[key] = be
            while len(self._backends) > 15:
                self._backends.popitem(last=False)[1].close()
        self._backends.move_to_end(key)
        self.initialized_backend = ("HIP", key)
        if self.cache_frame_transform:
            self.stab_data[timestamp_us] = self.get_frame_transform_at(pixel_type, timestamp_us, frame, buffers)

    def process_pixels(self, pixel_type, timestamp_us, frame, buffers, frame_transform=None):   # mod.rs:612-725
        if buffers.input.size[1] < 4 or buffers.output.size[1] < 4:
            raise GyroflowCoreError("SizeTooSmall")
        itm = frame_transform
        if itm is None:
            itm = self.stab_data.get(timestamp_us) if self.cache_frame_transform else self.get_frame_transform_at(pixel_type, timestamp_us, frame, buffers)
        if itm is None:
            raise GyroflowCoreError("NoStabilizationData", timestamp_us)
        kp = itm.kernel_params
        if self.size != (kp.width, kp.height):
            raise GyroflowCoreError("SizeMismatch", (self.size, (kp.width, kp.height)))
        if self.output_size != (kp.output_width, kp.output_height):
            raise GyroflowCoreError("SizeMismatch", (self.size, (kp.output_width, kp.output_height)))
        if buffers.input.size[0] > kp.stride:
            raise GyroflowCoreError("InvalidStride", (kp.stride, buffers.input.size[0]))
        if buffers.output.size[0] > kp.output_stride:
            raise GyroflowCoreError("InvalidStride", (kp.output_stride, buffers.output.size[0]))
        if buffers.input.nbytes() == 0:
            raise GyroflowCoreError("InputBufferEmpty")
        if buffers.output.nbytes() == 0:
            raise GyroflowCoreError("OutputBufferEmpty")
        self.ensure_ready_for_processing(pixel_type, timestamp_us, frame, buffers)
        be = self._backends[self.initialized_backend[1]]
        try:
            be.undistort_image(buffers.to_abi(), kp, itm.matrices, itm.mesh_data or None)
        except warp.GfwError as e:
            raise GyroflowCoreError(_CODE_TO_KIND.get(e.code, "Unknown"), str(e)) from e
        return ProcessedInfo(itm.fov, itm.minimal_fov, itm.focal_length, "HIP:" + warp.last_backend())

    def close(self):
        for be in self._backends.values():
            be.close()
        self._backends.clear()
