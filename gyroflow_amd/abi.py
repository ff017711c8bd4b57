"""ctypes view of include/gfwarp.h (the C ABI of libgfwarp).

Nothing here computes pixels: it declares the structs/enums of the boundary and
loads the HIP-built shared library.  There is no CPU fallback — if
``libgfwarp.so`` is missing, :func:`load_library` raises.
"""
import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "libgfwarp.so")


class KernelParams(C.Structure):
    """Byte-exact mirror of ``KernelParams`` (stabilization/mod.rs:101-150), 368 B."""
    _pack_ = 4
    _fields_ = [
        ("width", C.c_int32), ("height", C.c_int32), ("stride", C.c_int32),
        ("output_width", C.c_int32), ("output_height", C.c_int32), ("output_stride", C.c_int32),
        ("matrix_count", C.c_int32), ("interpolation", C.c_int32), ("background_mode", C.c_int32),
        ("flags", C.c_int32), ("bytes_per_pixel", C.c_int32), ("pix_element_count", C.c_int32),
        ("background", C.c_float * 4), ("f", C.c_float * 2), ("c", C.c_float * 2), ("k", C.c_float * 12),
        ("fov", C.c_float), ("r_limit", C.c_float), ("lens_correction_amount", C.c_float),
        ("input_vertical_stretch", C.c_float), ("input_horizontal_stretch", C.c_float),
        ("background_margin", C.c_float), ("background_margin_feather", C.c_float),
        ("canvas_scale", C.c_float), ("input_rotation", C.c_float), ("output_rotation", C.c_float),
        ("translation2d", C.c_float * 2), ("translation3d", C.c_float * 4),
        ("source_rect", C.c_int32 * 4), ("output_rect", C.c_int32 * 4),
        ("digital_lens_params", C.c_float * 16), ("safe_area_rect", C.c_float * 4),
        ("max_pixel_value", C.c_float), ("distortion_model", C.c_int32), ("digital_lens", C.c_int32),
        ("pixel_value_limit", C.c_float), ("light_refraction_coefficient", C.c_float),
        ("plane_index", C.c_int32), ("reserved1", C.c_float), ("reserved2", C.c_float),
        ("ewa_coeffs_p", C.c_float * 4), ("ewa_coeffs_q", C.c_float * 4),
    ]

    def copy(self):
        out = KernelParams()
        C.memmove(C.byref(out), C.byref(self), C.sizeof(KernelParams))
        return out


assert C.sizeof(KernelParams) == 368

# KernelParamsFlags (stabilization/mod.rs:83-99)
FLAG_FIX_COLOR_RANGE = 1
FLAG_HAS_DIGITAL_LENS = 2
FLAG_FILL_WITH_BACKGROUND = 4
FLAG_DRAWING_ENABLED = 8
FLAG_HORIZONTAL_RS = 16
FLAG_HAS_SOURCE_RECT = 32
FLAG_HAS_OUTPUT_RECT = 64
FLAG_FRAMEBUFFER_INVERTED = 128
FLAG_HAS_IBIS_DATA = 256
FLAG_HAS_MESH_DATA = 512
FLAG_HAS_FPD_DATA = 1024
FLAG_ANY_UNDERWATER = 2048

# Interpolation (stabilization/mod.rs:25-34)
INTERP = {"Bilinear": 2, "Bicubic": 4, "Lanczos4": 8, "RobidouxSharp": 10, "Robidoux": 11,
          "Mitchell": 12, "CatmullRom": 13}

# distortion model ids (include/gfwarp.h)
MODELS = {"none": 0, "opencv_fisheye": 1, "opencv_standard": 2, "poly3": 3, "poly5": 4, "ptlens": 5,
          "insta360": 6, "sony": 7, "generic_polynomial": 8, "gopro": 9, "gopro_superview": 10,
          "gopro_hyperview": 11, "gopro_warp": 12, "digital_stretch": 13, "gopro6_superview": 14}

# PixelType implementors (pixel_formats.rs): name -> (id, numpy dtype, element count, default_max_value)
PIXEL_TYPES = {
    "Luma8": (0, "u1", 1, 255.0), "Luma16": (1, "<u2", 1, 65535.0), "RGB8": (2, "u1", 3, 255.0),
    "RGBA8": (3, "u1", 4, 255.0), "BGRA8": (4, "u1", 4, 255.0), "RGB16": (5, "<u2", 3, 65535.0),
    "RGBA16": (6, "<u2", 4, 65535.0), "AYUV16": (7, "<u2", 4, 65535.0), "RGBAf": (8, "<f4", 4, None),
    "RGBAf16": (9, "<f2", 4, None), "R32f": (10, "<f4", 1, None), "UV8": (11, "u1", 2, 255.0),
    "UV16": (12, "<u2", 2, 65535.0),
}

BUF_NONE, BUF_HOST, BUF_HIP_DEVICE = 0, 1, 2


class FrameTiming(C.Structure):
    """``gfw_frame_timing``: inputs of the device-side per-row matrix builder."""
    _fields_ = [("timestamp_ms", C.c_double), ("per_frame_time_offset_ms", C.c_double), ("frame_readout_time_ms", C.c_double),
                ("new_k", C.c_double * 9), ("video_rotation_deg", C.c_double), ("rows", C.c_int32), ("readout_dim", C.c_int32),
                ("framebuffer_inverted", C.c_int32), ("suppress_rotation", C.c_int32)]


class FrameStab(C.Structure):
    """``gfw_frame_stab``: file_metadata.camera_stab_data[frame] (IBIS/OIS splines) for the device matrix builder."""
    _fields_ = [("offset", C.c_double), ("sensor_size", C.c_double * 2), ("crop_area", C.c_double * 4), ("pixel_pitch", C.c_double * 2),
                ("width", C.c_double), ("height", C.c_double), ("ibis_count", C.c_int32), ("ois_count", C.c_int32),
                ("ibis", C.c_void_p), ("ois", C.c_void_p)]


class BufferDesc(C.Structure):
    """``BufferDescription`` (gpu/mod.rs:17-24)."""
    _fields_ = [
        ("width", C.c_int32), ("height", C.c_int32), ("stride", C.c_int32),
        ("has_rect", C.c_int32), ("rect", C.c_int32 * 4),
        ("has_rotation", C.c_int32), ("rotation", C.c_float),
        ("kind", C.c_int32), ("texture_copy", C.c_int32),
        ("data", C.c_void_p), ("len", C.c_size_t),
    ]


class Buffers(C.Structure):
    """``Buffers`` (gpu/mod.rs:25-28)."""
    _fields_ = [("input", BufferDesc), ("output", BufferDesc)]


ERRORS = {
    0: "Ok", -1: "SizeTooSmall", -2: "SizeMismatch", -3: "InvalidStride", -4: "NoStabilizationData",
    -5: "InputBufferEmpty", -6: "OutputBufferEmpty", -7: "UnsupportedBuffer", -8: "BufferSizeMismatch",
    -9: "InvalidArgument", -10: "NoDevice", -11: "HipError", -100: "Unknown",
}

ERR_INVALID_ARGUMENT, ERR_NO_DEVICE, ERR_HIP = -9, -10, -11
POINT_INDEX_SINGLE, POINT_INDEX_PER_POINT, POINT_INDEX_PER_ROW, POINT_INDEX_PER_COLUMN = 0, 1, 2, 3
OPT_SYNCHRONOUS, OPT_MATRICES_ON_DEVICE, OPT_KERNEL_VARIANT, OPT_PROFILE, OPT_TUNE_ROWS, OPT_TUNE_GRID, OPT_JIT, OPT_COALESCE_PLANES, OPT_COALESCE_FRAMES = 1, 2, 3, 4, 5, 6, 7, 8, 9
OPT_FRAME_SYNC = 10
CLIP_MAX = 16                    # GFW_CLIP_FRAMES_MAX: frames gfw_undistort_clip puts into one launch

_lib = None


def bind(lib):
    """Declare argtypes/restypes of every symbol in include/gfwarp.h."""
    vp, i32, sz = C.c_void_p, C.c_int, C.c_size_t
    lib.gfw_abi_version.restype = i32
    lib.gfw_list_devices.argtypes = [C.c_char_p, sz]; lib.gfw_list_devices.restype = i32
    lib.gfw_set_device.argtypes = [i32]; lib.gfw_set_device.restype = i32
    lib.gfw_get_info.argtypes = [C.c_char_p, sz]; lib.gfw_get_info.restype = i32
    lib.gfw_is_buffer_supported.argtypes = [C.POINTER(Buffers)]; lib.gfw_is_buffer_supported.restype = i32
    lib.gfw_create.argtypes = [C.POINTER(KernelParams), i32, i32, i32, C.POINTER(Buffers), sz]
    lib.gfw_create.restype = vp
    lib.gfw_destroy.argtypes = [vp]; lib.gfw_destroy.restype = None
    lib.gfw_undistort_image.argtypes = [vp, C.POINTER(Buffers), C.POINTER(KernelParams), vp, i32, vp, sz, vp, sz]
    lib.gfw_undistort_image.restype = i32
    lib.gfw_undistort_frame.argtypes = [vp, i32, C.POINTER(Buffers), C.POINTER(KernelParams), C.POINTER(i32), vp, i32, vp, sz]
    lib.gfw_undistort_frame.restype = i32
    lib.gfw_set_option.argtypes = [vp, i32, C.c_int64]; lib.gfw_set_option.restype = i32
    lib.gfw_get_stream.argtypes = [vp]; lib.gfw_get_stream.restype = vp
    lib.gfw_set_stream.argtypes = [vp, vp]; lib.gfw_set_stream.restype = i32
    lib.gfw_synchronize.argtypes = [vp]; lib.gfw_synchronize.restype = i32
    lib.gfw_flush.argtypes = [vp]; lib.gfw_flush.restype = i32
    lib.gfw_import_external_fd.argtypes = [i32, C.c_size_t, C.c_ulonglong, C.POINTER(vp), C.POINTER(vp)]; lib.gfw_import_external_fd.restype = i32
    lib.gfw_release_external.argtypes = [vp]; lib.gfw_release_external.restype = i32
    lib.gfw_last_backend.argtypes = [vp]; lib.gfw_last_backend.restype = C.c_char_p
    lib.gfw_get_profile.argtypes = [vp, C.POINTER(C.c_double), C.POINTER(C.c_int64), i32]; lib.gfw_get_profile.restype = i32
    lib.gfw_get_profile_frames.argtypes = [vp, C.POINTER(C.c_double), C.POINTER(C.c_int64), C.POINTER(C.c_int64), i32]; lib.gfw_get_profile_frames.restype = i32
    lib.gfw_undistort_clip.argtypes = [vp, i32, i32, C.POINTER(Buffers), C.POINTER(KernelParams), C.POINTER(i32), C.POINTER(vp), i32]
    lib.gfw_undistort_clip.restype = i32
    lib.gfw_debug_jit_compile.argtypes = [C.c_char_p, C.c_char_p, C.c_char_p, C.c_char_p, C.c_char_p, sz]; lib.gfw_debug_jit_compile.restype = C.c_long
    lib.gfw_jit_status.argtypes = [vp, C.POINTER(C.c_double), C.c_char_p, sz]; lib.gfw_jit_status.restype = i32
    lib.gfw_set_quaternion_tracks.argtypes = [vp, vp, vp, i32, vp, vp, i32]; lib.gfw_set_quaternion_tracks.restype = i32
    lib.gfw_build_matrices.argtypes = [vp, C.POINTER(FrameTiming), vp, C.POINTER(vp)]; lib.gfw_build_matrices.restype = i32
    lib.gfw_build_matrices_stab.argtypes = [vp, C.POINTER(FrameTiming), C.POINTER(FrameStab), vp, C.POINTER(vp)]; lib.gfw_build_matrices_stab.restype = i32
    lib.gfw_set_sync_offsets.argtypes = [vp, C.c_double, vp, vp, i32]; lib.gfw_set_sync_offsets.restype = i32
    lib.gfw_build_matrices_batch.argtypes = [vp, C.POINTER(FrameTiming), i32, C.POINTER(vp)]; lib.gfw_build_matrices_batch.restype = i32
    lib.gfw_stmap_undistort.argtypes = [vp, C.POINTER(KernelParams), vp, i32, vp, sz, i32, i32, vp, i32]; lib.gfw_stmap_undistort.restype = i32
    lib.gfw_undistort_points.argtypes = [vp, C.POINTER(KernelParams), vp, sz, i32, vp, i32, vp, i32, vp, sz, vp, i32]; lib.gfw_undistort_points.restype = i32
    lib.gfw_pack_matrices.argtypes = [vp, i32, vp]; lib.gfw_pack_matrices.restype = i32
    lib.gfw_checksum64.argtypes = [vp, vp, sz, vp]; lib.gfw_checksum64.restype = i32
    lib.gfw_set_frame_checksums.argtypes = [vp, vp, sz]; lib.gfw_set_frame_checksums.restype = i32
    lib.gfw_get_audit.argtypes = [vp, C.POINTER(C.c_ulonglong * 8), i32]; lib.gfw_get_audit.restype = i32
    lib.gfw_debug_math.argtypes = [i32, vp, vp, vp, sz]; lib.gfw_debug_math.restype = i32
    lib.gfw_debug_selftest.argtypes = [i32, C.c_ulonglong, C.c_ulonglong]; lib.gfw_debug_selftest.restype = C.c_longlong
    lib.gfw_last_error.restype = C.c_char_p
    lib.gfw_debug_source_id.argtypes = [C.c_char_p, sz]; lib.gfw_debug_source_id.restype = i32
    lib.gfw_debug_paired_launches.argtypes = [vp]; lib.gfw_debug_paired_launches.restype = C.c_longlong
    lib.gfw_debug_frames_per_launch.argtypes = [C.c_ulonglong, C.c_ulonglong, i32]; lib.gfw_debug_frames_per_launch.restype = i32
    lib.gfw_pixel_type_info.argtypes = [i32, C.POINTER(i32), C.POINTER(i32), C.POINTER(C.c_float)]
    lib.gfw_pixel_type_info.restype = i32
    return lib


EXPORTS = ["gfw_abi_version", "gfw_list_devices", "gfw_set_device", "gfw_get_info", "gfw_is_buffer_supported",
           "gfw_create", "gfw_destroy", "gfw_undistort_image", "gfw_undistort_frame", "gfw_set_option",
           "gfw_get_stream", "gfw_set_stream", "gfw_synchronize", "gfw_flush", "gfw_import_external_fd", "gfw_release_external", "gfw_last_backend", "gfw_get_profile", "gfw_last_error", "gfw_debug_math", "gfw_debug_jit_key", "gfw_debug_selftest", "gfw_get_audit", "gfw_pack_matrices", "gfw_checksum64", "gfw_set_frame_checksums", "gfw_set_quaternion_tracks", "gfw_build_matrices", "gfw_build_matrices_stab", "gfw_set_sync_offsets", "gfw_build_matrices_batch", "gfw_stmap_undistort", "gfw_undistort_points",
           "gfw_pixel_type_info", "gfw_undistort_clip", "gfw_jit_status", "gfw_get_profile_frames", "gfw_debug_jit_compile", "gfw_debug_source_id", "gfw_debug_p1_radial", "gfw_debug_paired_launches", "gfw_debug_frames_per_launch"]


def load_library(path=None):
    """Load libgfwarp.so (built by ``__graft_entry__.build()``).  Fails loudly when absent."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    p = path or os.environ.get("GFW_LIBRARY", "") or LIB_PATH      # GFW_LIBRARY: A/B builds of the library (tools/, benchmarking)
    if not os.path.exists(p):
        raise RuntimeError(
            "libgfwarp.so not found at %s — the HIP extension is not built. Run "
            "`python -c 'import __graft_entry__ as g; g.build()'`. There is no CPU fallback." % p)
    lib = bind(C.CDLL(p))
    if path is None:
        _lib = lib
    return lib


def kernel_source_id():
    """identity of the fused kernel's source inside the loaded library (gfw_debug_source_id): what a stored measurement must name to be quoted for it"""
    buf = C.create_string_buffer(128)
    n = load_library().gfw_debug_source_id(buf, len(buf))
    return buf.value.decode() if n > 0 else ""
