/* gfwarp.h — C ABI of libgfwarp, the MI355X (gfx950) warp backend.
 *
 * This is the drop-in boundary for gyroflow-core's per-pixel
 * undistort -> rotate (rolling shutter) -> redistort -> sample path.  Every
 * entry point states the reference interface it stands in for
 * (paths relative to the gyroflow tree, v1.6.3):
 *
 *   backend object          src/core/gpu/opencl.rs:178   OclWrapper::new
 *                           src/core/gpu/wgpu.rs:147     WgpuWrapper::new
 *   per-plane call          src/core/gpu/opencl.rs:330   OclWrapper::undistort_image
 *                           src/core/gpu/wgpu.rs:454     WgpuWrapper::undistort_image
 *   CPU twin (authoritative arithmetic)
 *                           src/core/stabilization/cpu_undistort.rs:233
 *   uniform block           src/core/stabilization/mod.rs:101-150  KernelParams
 *   buffers                 src/core/gpu/mod.rs:17-71    Buffers / BufferDescription / BufferSource
 *
 * Plain C: pointers and sizes only, no C++/torch types.  All functions are
 * thread-safe with respect to *different* contexts; one context is
 * single-thread-affine exactly like the reference's thread-local backend
 * caches (stabilization/mod.rs:59-66).
 */
#ifndef GFWARP_H
#define GFWARP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GFW_ABI_VERSION 2

/* ---- KernelParams: byte-exact mirror of stabilization/mod.rs:101-150 ------
 * #[repr(C, packed(4))], 92 four-byte words = 368 bytes. */
typedef struct gfw_kernel_params {
    int32_t width;               /*   0 */
    int32_t height;              /*   4 */
    int32_t stride;              /*   8  bytes */
    int32_t output_width;        /*  12 */
    int32_t output_height;       /*  16 */
    int32_t output_stride;       /*  20  bytes */
    int32_t matrix_count;        /*  24  1 = no rolling-shutter correction */
    int32_t interpolation;       /*  28  2,4,8 = LUT taps; 10..13 = EWA */
    int32_t background_mode;     /*  32  0 solid,1 repeat,2 mirror,3 margin+feather */
    int32_t flags;               /*  36  GFW_FLAG_* */
    int32_t bytes_per_pixel;     /*  40 */
    int32_t pix_element_count;   /*  44 */
    float   background[4];       /*  48 */
    float   f[2];                /*  64 */
    float   c[2];                /*  72 */
    float   k[12];               /*  80 */
    float   fov;                 /* 128 */
    float   r_limit;             /* 132 */
    float   lens_correction_amount;    /* 136 */
    float   input_vertical_stretch;    /* 140 */
    float   input_horizontal_stretch;  /* 144 */
    float   background_margin;         /* 148 */
    float   background_margin_feather; /* 152 */
    float   canvas_scale;              /* 156 */
    float   input_rotation;            /* 160  degrees */
    float   output_rotation;           /* 164  degrees */
    float   translation2d[2];          /* 168 */
    float   translation3d[4];          /* 176 */
    int32_t source_rect[4];            /* 192  x,y,w,h */
    int32_t output_rect[4];            /* 208  x,y,w,h */
    float   digital_lens_params[16];   /* 224 */
    float   safe_area_rect[4];         /* 288 */
    float   max_pixel_value;           /* 304 */
    int32_t distortion_model;          /* 308  GFW_MODEL_* (not trusted: see gfw_create) */
    int32_t digital_lens;              /* 312 */
    float   pixel_value_limit;         /* 316 */
    float   light_refraction_coefficient; /* 320 */
    int32_t plane_index;               /* 324 */
    float   reserved1;                 /* 328 */
    float   reserved2;                 /* 332 */
    float   ewa_coeffs_p[4];           /* 336 */
    float   ewa_coeffs_q[4];           /* 352 */
} gfw_kernel_params;                   /* 368 */

/* KernelParamsFlags, stabilization/mod.rs:83-99 */
enum {
    GFW_FLAG_FIX_COLOR_RANGE      = 1 << 0,
    GFW_FLAG_HAS_DIGITAL_LENS     = 1 << 1,
    GFW_FLAG_FILL_WITH_BACKGROUND = 1 << 2,
    GFW_FLAG_DRAWING_ENABLED      = 1 << 3,
    GFW_FLAG_HORIZONTAL_RS        = 1 << 4,
    GFW_FLAG_HAS_SOURCE_RECT      = 1 << 5,
    GFW_FLAG_HAS_OUTPUT_RECT      = 1 << 6,
    GFW_FLAG_FRAMEBUFFER_INVERTED = 1 << 7,
    GFW_FLAG_HAS_IBIS_DATA        = 1 << 8,
    GFW_FLAG_HAS_MESH_DATA        = 1 << 9,
    GFW_FLAG_HAS_FPD_DATA         = 1 << 10,
    GFW_FLAG_ANY_UNDERWATER       = 1 << 11
};

/* Interpolation, stabilization/mod.rs:25-34 */
enum {
    GFW_INTERP_BILINEAR       = 2,
    GFW_INTERP_BICUBIC        = 4,
    GFW_INTERP_LANCZOS4       = 8,
    GFW_INTERP_ROBIDOUX_SHARP = 10,
    GFW_INTERP_ROBIDOUX       = 11,
    GFW_INTERP_MITCHELL       = 12,
    GFW_INTERP_CATMULL_ROM    = 13
};

/* Distortion model ids: src/core/gpu/stabilize_spirv/src/distortion_models/mod.rs:62-81
 * (declaration order of impl_models!).  GoPro6Superview has no id there
 * (it exists only in stabilization/distortion_models/mod.rs:107); 14 is ours. */
enum {
    GFW_MODEL_NONE               = 0,
    GFW_MODEL_OPENCV_FISHEYE     = 1,
    GFW_MODEL_OPENCV_STANDARD    = 2,
    GFW_MODEL_POLY3              = 3,
    GFW_MODEL_POLY5              = 4,
    GFW_MODEL_PTLENS             = 5,
    GFW_MODEL_INSTA360           = 6,
    GFW_MODEL_SONY               = 7,
    GFW_MODEL_GENERIC_POLYNOMIAL = 8,
    GFW_MODEL_GOPRO              = 9,
    GFW_MODEL_GOPRO_SUPERVIEW    = 10,
    GFW_MODEL_GOPRO_HYPERVIEW    = 11,
    GFW_MODEL_GOPRO_WARP         = 12,
    GFW_MODEL_DIGITAL_STRETCH    = 13,
    GFW_MODEL_GOPRO6_SUPERVIEW   = 14
};

/* PixelType implementors, stabilization/pixel_formats.rs:48-60 (declaration order) */
enum {
    GFW_PIX_LUMA8   = 0,
    GFW_PIX_LUMA16  = 1,
    GFW_PIX_RGB8    = 2,
    GFW_PIX_RGBA8   = 3,
    GFW_PIX_BGRA8   = 4,
    GFW_PIX_RGB16   = 5,
    GFW_PIX_RGBA16  = 6,
    GFW_PIX_AYUV16  = 7,
    GFW_PIX_RGBAF   = 8,
    GFW_PIX_RGBAF16 = 9,
    GFW_PIX_R32F    = 10,
    GFW_PIX_UV8     = 11,
    GFW_PIX_UV16    = 12,
    GFW_PIX_COUNT   = 13
};

/* BufferSource, gpu/mod.rs:29-71.  HOST = BufferSource::Cpu{buffer};
 * HIP_DEVICE = the analogue of CUDABuffer{buffer} (a device pointer that
 * already lives in this GPU's HBM). */
enum {
    GFW_BUF_NONE       = 0,
    GFW_BUF_HOST       = 1,
    GFW_BUF_HIP_DEVICE = 2
};

/* BufferDescription, gpu/mod.rs:17-24 */
typedef struct gfw_buffer_desc {
    int32_t width, height, stride;   /* size: (w, h, stride in bytes) */
    int32_t has_rect;                /* Option<(x,y,w,h)> */
    int32_t rect[4];
    int32_t has_rotation;            /* Option<f32>, degrees */
    float   rotation;
    int32_t kind;                    /* GFW_BUF_* */
    int32_t texture_copy;            /* kept for layout parity; unused */
    void   *data;
    size_t  len;                     /* bytes available at data */
} gfw_buffer_desc;

/* Buffers, gpu/mod.rs:25-28 */
typedef struct gfw_buffers {
    gfw_buffer_desc input;
    gfw_buffer_desc output;
} gfw_buffers;

/* Error codes: GyroflowCoreError (src/core/lib.rs:2099-2141) + backend-level
 * conditions that the reference logs-and-skips (opencl.rs:336-358). */
enum {
    GFW_OK                        =  0,
    GFW_ERR_SIZE_TOO_SMALL        = -1,   /* SizeTooSmall          mod.rs:613 */
    GFW_ERR_SIZE_MISMATCH         = -2,   /* SizeMismatch          mod.rs:636-637 */
    GFW_ERR_INVALID_STRIDE        = -3,   /* InvalidStride         mod.rs:639-640 */
    GFW_ERR_NO_STABILIZATION_DATA = -4,   /* NoStabilizationData   mod.rs:721 */
    GFW_ERR_INPUT_BUFFER_EMPTY    = -5,   /* InputBufferEmpty      lib.rs:890 */
    GFW_ERR_OUTPUT_BUFFER_EMPTY   = -6,   /* OutputBufferEmpty     lib.rs:891 */
    GFW_ERR_UNSUPPORTED_BUFFER    = -7,   /* is_buffer_supported == false */
    GFW_ERR_BUFFER_SIZE_MISMATCH  = -8,   /* "Buffer size mismatch" opencl.rs:336-358 */
    GFW_ERR_INVALID_ARGUMENT      = -9,
    GFW_ERR_NO_DEVICE             = -10,  /* no gfx950 device / HIP runtime error */
    GFW_ERR_HIP                   = -11,
    GFW_ERR_UNKNOWN               = -100  /* Unknown */
};

typedef struct gfw_ctx gfw_ctx;

/* ---- device management --------------------------------------------------
 * OclWrapper::list_devices opencl.rs:60, ::set_device :93,
 * ::initialize_context :118, ::get_info (wgpu.rs:123). */
int  gfw_abi_version(void);
/* Writes '\n'-separated device names ("[HIP] <name>") into buf; returns the
 * device count, or a negative GFW_ERR_* */
int  gfw_list_devices(char *buf, size_t cap);
int  gfw_set_device(int index);
int  gfw_get_info(char *buf, size_t cap);
/* is_buffer_supported(): opencl.rs:451 / wgpu.rs:562 */
int  gfw_is_buffer_supported(const gfw_buffers *buffers);

/* ---- backend object -----------------------------------------------------
 * OclWrapper::new(params, ocl_names, distortion_model, digital_lens,
 *                 buffers, drawing_len)              opencl.rs:178
 * The model ids are explicit arguments because FrameTransform leaves
 * KernelParams.distortion_model/digital_lens at their defaults
 * (frame_transform.rs:322-340).  digital_lens = GFW_MODEL_NONE for "None".
 * Owns: the stream, device staging for HOST buffers (sized from `buffers`),
 * matrices (14 * (flags&16 ? width : height) floats), params, mesh
 * (MAX_BUFFER_SIZE = 839 floats, gyro_source/splines.rs:88-89).
 * Returns NULL on failure; see gfw_last_error(). */
gfw_ctx *gfw_create(const gfw_kernel_params *params, int pixel_type,
                    int distortion_model, int digital_lens,
                    const gfw_buffers *buffers, size_t drawing_len);
void     gfw_destroy(gfw_ctx *ctx);

/* OclWrapper::undistort_image(&self, buffers, itm: &FrameTransform, drawing)
 *                                                             opencl.rs:330
 * `matrices` = FrameTransform.matrices flattened ([f32;14] per row,
 * frame_transform.rs:13); host pointer unless GFW_MATRICES_ON_DEVICE is set
 * with gfw_set_option.  Synchronous for HOST buffers (output complete on
 * return, like opencl.rs:413); for HIP_DEVICE buffers the work is enqueued on
 * the context stream and the call returns after enqueue unless the context
 * is in synchronous mode (default: synchronous).
 * Returns GFW_OK or a negative code; never aborts. */
int gfw_undistort_image(gfw_ctx *ctx, const gfw_buffers *buffers,
                        const gfw_kernel_params *params,
                        const float *matrices, int matrix_count,
                        const uint8_t *drawing, size_t drawing_len,
                        const float *mesh, size_t mesh_len);

/* Additive: all planes of one frame in one launch, sharing the per-row
 * matrices and the coordinate evaluation between planes (the reference runs
 * one process_pixels per plane: src/rendering/mod.rs:655-658).  planes[i] /
 * params[i] / pixel_types[i] describe plane i exactly as the i-th
 * gfw_undistort_image call would.  Results are bit-identical to calling
 * gfw_undistort_image once per plane. */
int gfw_undistort_frame(gfw_ctx *ctx, int nplanes,
                        const gfw_buffers *planes,
                        const gfw_kernel_params *params,
                        const int *pixel_types,
                        const float *matrices, int matrix_count,
                        const float *mesh, size_t mesh_len);
/* The frame loop of a render on the library side (the reference drives process_pixels once per frame from
 * rendering/mod.rs:487-547): n_frames frames of one clip — `planes` holds n_frames * nplanes descriptions, frame-major;
 * `params` (nplanes entries) and `pixel_types` are shared by all frames; matrices[f] is frame f's table, with the meaning
 * GFW_OPT_MATRICES_ON_DEVICE gives it.  Frames are warped in order on the context's stream with the results of
 * gfw_undistort_frame; frames with HIP_DEVICE buffers and device-resident tables that share the context's run-time specialised
 * kernel (GFW_OPT_JIT) leave in launches of up to GFW_CLIP_FRAMES_MAX frames, so that the occupancy tail of one frame is
 * filled by the next and the cost differences between image regions average out over the GPU's partitions.  The frames of
 * one launch are in flight together; a frame whose buffers overlap a pending frame's (it writes or reads a destination
 * already in the launch, or writes one of its sources) starts a new launch, so the results are those of the ordered calls.
 * A launch also takes at most 1.1 GB of source + destination (sixteen 4K 16-bit 4:2:2 frames, four 8K ones; at least two
 * frames; environment GFW_CLIP_LAUNCH_MB overrides): past that the frames in flight together cost more than the tail they
 * fill (8K: 9 %), and the call's frames are dealt evenly over the launches it needs (16 under a cap of 4: 4 + 4 + 4 + 4). */
#define GFW_CLIP_FRAMES_MAX 16
int gfw_undistort_clip(gfw_ctx *ctx, int n_frames, int nplanes,
                       const gfw_buffers *planes,
                       const gfw_kernel_params *params,
                       const int *pixel_types,
                       const float *const *matrices, int matrix_count);

/* ---- options / stream ---------------------------------------------------*/
enum {
    GFW_OPT_SYNCHRONOUS        = 1,  /* 1 (default): return after stream sync */
    GFW_OPT_MATRICES_ON_DEVICE = 2,  /* 0: host rows of [f32;14] (default, uploaded per call like opencl.rs:406);
                                        1: device pointer, rows of [f32;14]; 2: device pointer, rows of 16 floats as
                                        written by gfw_pack_matrices (no per-call work at all).  With 1 and 2 the
                                        library cannot scan the rows, so the IBIS/OIS terms m[9..13] are honoured only if
                                        GFW_FLAG_HAS_IBIS_DATA is set in params->flags (get_kernel_flags, mod.rs:226-251,
                                        sets it for every clip that has such data); with 1 the roll's cos/sin are
                                        evaluated on the device with the host libm's own routines (gfw_math.h). */
    GFW_OPT_KERNEL_VARIANT     = 3,  /* (further values: gfwarp_testing.h)  0 auto; 1 generic per-plane kernel; 2 fused kernel with the exact first pass;
                                        3 fused kernel, certified first pass in audit mode (see gfw_get_audit); 4 the same with the first pass evaluated
                                        per pixel (the form of rounds 2-4; by default a frame takes the lattice form where its certificate allows).
                                        Every variant produces the same pixels; any other value is rejected (GFW_ERR_INVALID_ARGUMENT) */
    GFW_OPT_PROFILE            = 4,  /* 1: bracket every warp-kernel launch with hipEvents on the context stream */
    GFW_OPT_TUNE_ROWS          = 5,  /* reserved (ignored) */
    GFW_OPT_TUNE_GRID          = 6,  /* tuning: persistent workgroups of the fused kernel (0 = 6 per CU) */
    GFW_OPT_JIT                = 7,  /* per-clip specialised kernel, compiled at run time the way the reference compiles its OpenCL source per clip
                                        (opencl.rs:181-214): 0 never; 1 (default) built in the background once the context has warped three
                                        frames with the same clip constants, frames run ahead-of-time until it is ready; 2 built at the first
                                        frame, which waits for it (~1 s).  Same results bit for bit; the environment variable GFW_JIT sets
                                        the default of new contexts.  Without libhiprtc.so the option has no effect. */
    GFW_OPT_COALESCE_PLANES    = 8,  /* the planes of a frame that arrive one per gfw_undistort_image call — the reference's render loop issues
                                        process_pixels once per plane, each plane through its own Stabilization / backend object
                                        (src/rendering/mod.rs:494-545) — leave as ONE fused launch.  Applies to calls that are stream-ordered anyway:
                                        GFW_OPT_SYNCHRONOUS = 0 (or GFW_OPT_FRAME_SYNC), HIP_DEVICE buffers on both sides.  Such a call validates its
                                        arguments, keeps a copy and returns; the frame is enqueued on the stream of the context that took plane_index 0
                                        when its last plane arrives — a UV8 / UV16 plane, the third Luma plane, the fourth R32f plane — or when anything
                                        else is asked of a member context (gfw_synchronize, gfw_flush, gfw_get_stream, an option, a plane that does not
                                        continue the frame, gfw_destroy).
                                        1 (default): only calls on contexts that have been SEEN as planes of a multi-plane frame are held (a call with
                                        plane_index k + 1 following, on the same thread, a call with plane_index k on another context of the same device,
                                        lens and matrix count): a clip's first frame leaves plane by plane, a lone greyscale / float plane is never held.
                                        2: every eligible call is held from the first frame on.  0: every call launches its own plane.
                                        ORDERING CONTRACT: the fused launch waits for everything that was enqueued on EACH member context's stream before
                                        the frame was completed (a plane's upload or decode, a consumer still reading its destination), and every member's
                                        stream is ordered behind the launch.  Observe completion through gfw_synchronize / gfw_flush / gfw_get_stream of
                                        any member context (or events recorded after them).  Results are bit-identical to the per-plane launches. */
    GFW_OPT_COALESCE_FRAMES    = 9,  /* frames assembled by GFW_OPT_COALESCE_PLANES that are held for one launch of the run-time specialised kernel
                                        (1..GFW_CLIP_FRAMES_MAX; default 1: a frame leaves when it is complete).  Larger values trade latency for the
                                        throughput of gfw_undistort_clip (the occupancy tail of one frame filled by the next).  Ignored by a
                                        synchronous owner (GFW_OPT_FRAME_SYNC). */
    GFW_OPT_FRAME_SYNC         = 10  /* 0 (default).  1: on a SYNCHRONOUS context (GFW_OPT_SYNCHRONOUS = 1, the reference's contract: opencl.rs:413) with
                                        HIP_DEVICE buffers, "complete on return" is relaxed to "the FRAME is complete when its LAST plane's call returns":
                                        the calls of the frame's earlier planes validate, are held (GFW_OPT_COALESCE_PLANES) and return at once, the last
                                        plane's call launches the fused kernel and waits for it.  This is what the render loop needs — it consumes a
                                        frame's planes only after all of its process_pixels calls (rendering/mod.rs:494-545) — but it is NOT what a caller
                                        that reads plane 0 right after plane 0's call gets; hence opt-in, on every context of the frame. */
};
int   gfw_set_option(gfw_ctx *ctx, int option, int64_t value);
/* Enqueues whatever gfw_undistort_image calls GFW_OPT_COALESCE_PLANES / _FRAMES are holding for a frame or launch this context belongs to
 * (no-op otherwise).  Does not wait for the GPU: gfw_synchronize does both. */
int   gfw_flush(gfw_ctx *ctx);
/* hipStream_t the context enqueues on (as void*); caller may substitute its
 * own stream (e.g. the decoder's) — mirrors passing the cl_command_queue in
 * BufferSource::OpenCL{queue} (gpu/mod.rs:37-40). */
void *gfw_get_stream(gfw_ctx *ctx);
/* Drains the stream in use (hipStreamSynchronize), then adopts `hip_stream` (not owned; NULL = the legacy default stream). */
int   gfw_set_stream(gfw_ctx *ctx, void *hip_stream);
int   gfw_synchronize(gfw_ctx *ctx);
/* Name of the kernel path the last call took ("plane_generic", "yuv_fused", ...):
 * the analogue of ProcessedInfo.backend (stabilization/mod.rs:194-200). */
const char *gfw_last_backend(gfw_ctx *ctx);
/* State of the context's run-time specialised kernel: 0 none / not available, 1 compiling, 2 ready (in use), 3 failed (frames keep
 * running ahead-of-time).  compile_ms and the compiler log (truncated to cap) are optional outputs. */
int   gfw_jit_status(gfw_ctx *ctx, double *compile_ms, char *log, size_t cap);

/* With GFW_OPT_PROFILE on: accumulated warp-kernel time (ms, hipEventElapsedTime on the context stream)
 * and launch count since the last reset; synchronises the stream.  reset != 0 clears the accumulators. */
int   gfw_get_profile(gfw_ctx *ctx, double *kernel_ms, int64_t *launches, int reset);
/* the same, plus the number of frames the bracketed launches covered (a gfw_undistort_clip launch carries up to GFW_CLIP_FRAMES_MAX) */
int   gfw_get_profile_frames(gfw_ctx *ctx, double *kernel_ms, int64_t *launches, int64_t *frames, int reset);

/* Thread-local, human-readable description of the last failure. */
const char *gfw_last_error(void);

/* Host helper: FrameTransform.matrices rows ([f32;14]) -> libgfwarp's 64-byte device row layout
 * (m0..m13, cosf(-m11), sinf(-m11) evaluated with the host libm as cpu_undistort.rs:159-160 does).  A pipeline that
 * knows its FrameTransforms ahead of time packs them once, keeps them in HBM and passes the device pointer with
 * GFW_OPT_MATRICES_ON_DEVICE = 2. */
int   gfw_pack_matrices(const float *rows14, int count, float *rows16);

/* ---- decoder / interop surfaces without the host ("next" row f-4) ----
 * The reference's zero-copy path imports memory another API owns and maps it to a device pointer of the compute API
 * (src/core/gpu/wgpu_interop_cuda.rs:181-215: cuImportExternalMemory -> cuExternalMemoryGetMappedBuffer; plane descriptors
 * src/rendering/zero_copy.rs:67-112; BufferSource::CUDABuffer, src/core/gpu/mod.rs:67-70).  gfw_import_external_fd does the same for a POSIX
 * file descriptor that names a device allocation (a dma-buf exported by the decoder / VA-API / Vulkan, or another process's
 * hipMemExportToShareableHandle): `*dev_ptr_out` is a device pointer to `size` bytes, to be used in GFW_BUF_HIP_DEVICE buffer descriptions —
 * plane offsets and the row pitch are the caller's (`data + offset`, `stride`).  `drm_format_modifier` states the surface layout: only
 * DRM_FORMAT_MOD_LINEAR (0) can be warped in place, anything else returns GFW_ERR_UNSUPPORTED_BUFFER.  The caller keeps its fd.
 * gfw_release_external unmaps (after the device has drained). */
typedef struct gfw_external gfw_external;
int   gfw_import_external_fd(int fd, size_t size, unsigned long long drm_format_modifier, void **dev_ptr_out, gfw_external **handle_out);
int   gfw_release_external(gfw_external *handle);

/* Verification helper for frame-sharded clip runs (no reference counterpart; SURVEY.md 8e: one 8-byte checksum per frame,
 * all-gathered across ranks): adds the sum of the u64 words of a device buffer (mod 2^64) to *d_out, enqueued in order
 * on the context's stream.  `bytes` a multiple of 8, `d_buf` 16-byte aligned, `d_out` a zero-initialised device word. */
int   gfw_checksum64(gfw_ctx *ctx, const void *d_buf, size_t bytes, unsigned long long *d_out);
/* The same checksum taken WHERE THE PIXELS LEAVE: from this call on, frame k submitted on the context (k = 0, 1, ...: a gfw_undistort_frame call, a frame of
 * gfw_undistort_clip, a frame assembled from coalesced gfw_undistort_image calls, a lone gfw_undistort_image call) adds the checksum of every byte it WRITES —
 * the byte times 256^(its address mod 8), modulo 2^64: what gfw_checksum64 of a zero-initialised destination holds afterwards — to d_sums[k mod count], in
 * order on the context's stream.  `d_sums`: `count` device words the caller zeroed; NULL or count = 0 turns it off.  A specialised fused kernel takes the sum in
 * its store path (no second pass over the output: 33 MB per C2 frame, 11 us); every other kernel is followed by a pass over the written region — same value.
 * HIP_DEVICE outputs only (GFW_ERR_INVALID_ARGUMENT from the frame's call otherwise). */
int   gfw_set_frame_checksums(gfw_ctx *ctx, unsigned long long *d_sums, size_t count);

/* ---- per-row matrices on the device ("next" row: FrameTransform::at_timestamp, frame_transform.rs:221-308) ----
 * gfw_set_quaternion_tracks uploads the clip's original and smoothed orientation tracks once
 * (GyroSource.quaternions / smoothed_quaternions: BTreeMap<i64 timestamp_us, Quat64>, gyro_source/mod.rs:857-882);
 * gfw_build_matrices then produces the packed rows of one frame directly in HBM (device pointer `rows16_out`, or a
 * context-owned table when NULL; its address is returned through `*out_ptr`) to be passed to
 * gfw_undistort_image/frame with GFW_OPT_MATRICES_ON_DEVICE = 2.  Results equal the host f64 statement to ~1 ULP of f32
 * (SVD vs closed-form inverse).  gfw_set_sync_offsets adds the clip's gyro/video sync offsets, gfw_build_matrices_stab the
 * IBIS/OIS terms, gfw_frame_timing.suppress_rotation the `suppress_rotation` switch: with them FrameTransform::at_timestamp's
 * matrix loop (frame_transform.rs:249-308) is covered in full. */
typedef struct gfw_frame_timing {
    double timestamp_ms;               /* frame centre */
    double per_frame_time_offset_ms;   /* file_metadata.per_frame_time_offsets[frame] */
    double frame_readout_time_ms;      /* signed, as get_frame_readout_time returns it */
    double new_k[9];                   /* get_new_k(), row-major (frame_transform.rs:37-51) */
    double video_rotation_deg;
    int32_t rows;                      /* matrix_count: H, W (horizontal readout) or 1 */
    int32_t readout_dim;               /* divisor of the row readout time: height, or width for horizontal readout */
    int32_t framebuffer_inverted;
    int32_t suppress_rotation;         /* 0; 1 = params.suppress_rotation (R = identity, frame_transform.rs:291-296);
                                          2 = the same with params.frame_readout_time == 0.0 (the IBIS/OIS terms are zeroed too).
                                          MUST be set (zero-initialise the struct): until GFW_ABI_VERSION 1 this slot was padding;
                                          any other value is rejected with GFW_ERR_INVALID_ARGUMENT */
} gfw_frame_timing;
/* file_metadata.camera_stab_data[frame] (gyro_source/file_metadata.rs:41-48): in-body / optical stabiliser positions along the
 * sensor readout, as two Catmull-Rom splines of the sensor row.  frame_transform.rs:234-241 and :270-289 turn them into
 * the per-row terms m[9..13] = (sx, sy, roll angle, ox, oy); gfw_build_matrices_stab evaluates them per row on the device
 * (f64, the reference's operation order) and fills the roll's cos/sin slots with the host libm's routines (gfw_math.h). */
typedef struct gfw_frame_stab {
    double offset;                     /* CameraStabData.offset */
    double sensor_size[2];             /* (u32, u32) */
    double crop_area[4];               /* (f32 x, y, w, h) widened exactly */
    double pixel_pitch[2];             /* (u32, u32) */
    double width, height;              /* params.width, params.height */
    int32_t ibis_count, ois_count;     /* control points of the two splines (ascending positions) */
    const double *ibis;                /* host, ibis_count x 4: position, x, y, z */
    const double *ois;                 /* host, ois_count x 4: position, x, y, z (z unused) */
} gfw_frame_stab;
int   gfw_set_quaternion_tracks(gfw_ctx *ctx, const int64_t *org_ts_us, const double *org_wxyz, int org_count,
                                const int64_t *smoothed_ts_us, const double *smoothed_wxyz, int smoothed_count);
int   gfw_build_matrices(gfw_ctx *ctx, const gfw_frame_timing *timing, float *rows16_out, float **out_ptr);
/* The same with the frame's IBIS/OIS stabiliser data (`stab` may be NULL): rows carry m[9..13] and cos/sin(-m[11]). */
int   gfw_build_matrices_stab(gfw_ctx *ctx, const gfw_frame_timing *timing, const gfw_frame_stab *stab, float *rows16_out, float **out_ptr);
/* Gyro/video synchronisation of the clip (GyroSource.offsets_adjusted: BTreeMap<i64 timestamp_us, f64 offset_ms> and
 * duration_ms): quat_at_timestamp subtracts offset_at_video_timestamp(t) from every lookup time — per row, since the offset is
 * interpolated at the row's own time — and returns identity when duration_ms <= 0 (gyro_source/mod.rs:857-860, :884-908).
 * Not calling this leaves no offsets and a positive duration.  `count` may be 0. */
int   gfw_set_sync_offsets(gfw_ctx *ctx, double duration_ms, const int64_t *timestamps_us, const double *offsets_ms, int count);
/* The tables of `count` (<= 64) upcoming frames in one launch, in order on the context's stream, into context-owned
 * memory (two batches alternate: a batch stays valid until the second next call).  out_ptrs[i] = device table of frame i,
 * to be passed as `matrices` with GFW_OPT_MATRICES_ON_DEVICE = 2.  Amortises the builder's latency and needs no
 * cross-stream synchronisation: the per-frame cost in a render loop drops to ~1 us of GPU time. */
int   gfw_build_matrices_batch(gfw_ctx *ctx, const gfw_frame_timing *timings, int count, float **out_ptrs);

/* ---- STMap coordinate export ("next" row: src/core/stmap.rs:87-109, :127-137) ----------------------------
 * The "undist" ST map: for every pixel (x, y) of a width x height map, the rolling-shutter row pick followed by
 * rotate_and_distort, written as two f32 (source x, y in pixels) instead of being sampled.  `params` is the
 * KernelParams stmap.rs builds (width/height/output_* = map size, flags = HAS_DIGITAL_LENS | HORIZONTAL_RS).
 * Pixels whose projection is None are left untouched (parallel_exr leaves them 0).  Bit-exact vs the CPU closure. */
int   gfw_stmap_undistort(gfw_ctx *ctx, const gfw_kernel_params *params, const float *matrices, int matrix_count,
                          const float *mesh, size_t mesh_len, int width, int height, float *coords, int coords_on_device);

/* ---- inverse point map ("next" row 3: cpu_undistort.rs:652-858 `undistort_points`; stmap.rs:123-127 "dist") ----
 * Source-image points -> stabilised output coordinates.  The STMap "dist" pass and the optical-flow caller
 * (cpu_undistort.rs:643-650) run it with lens_correction_amount == 1; with params->lens_correction_amount < 1 the
 * Newton inverse of the render's lens-correction blend follows (:785-851, what the zoom search uses), with
 * params->fov as its `fov`.  The lens / digital-lens models are the ones given to gfw_create.
 *   params     the KernelParams undistort_points builds (:669-681: width/height/output_*, f, c, k,
 *              digital_lens_params, light_refraction_coefficient) with input_*_stretch = lens.input_*_stretch (:704-705)
 *   points     n x 2 f32 (host), or NULL = the pixel grid parallel_exr walks: point i = (i % grid_width, i / grid_width)
 *   rotations  [rotation_count][9] row-major f32 = nalgebra::convert::<f64,f32> of `new_k * R` per point
 *              (FrameTransform::at_timestamp_for_points, frame_transform.rs:391-410)
 *   shifts     NULL, or [rotation_count][5] = (sx, sy, angle_rad, ox, oy) (frame_transform.rs:412-440)
 *   index_mode which rotation/shift row a point uses: GFW_POINT_INDEX_SINGLE (row 0: no rolling shutter),
 *              _PER_POINT (row i), _PER_ROW (grid y; vertical rolling shutter), _PER_COLUMN (grid x; horizontal);
 *              an index past rotation_count falls back to row 0 as `rot_per_point.get(index).unwrap_or(&rr)` does
 *   mesh       NULL, or the f64 lens mesh `undistort_points` receives (:714-753)
 *   out        n x 2 f32, host or device memory (out_on_device); (-1e6, -1e6) where the lens inverse is None (:855)
 * Bit-exact vs the CPU statement.  Synchronous. */
enum { GFW_POINT_INDEX_SINGLE = 0, GFW_POINT_INDEX_PER_POINT = 1, GFW_POINT_INDEX_PER_ROW = 2, GFW_POINT_INDEX_PER_COLUMN = 3 };
int   gfw_undistort_points(gfw_ctx *ctx, const gfw_kernel_params *params, const float *points, size_t n, int grid_width,
                           const float *rotations, int rotation_count, const float *shifts, int index_mode,
                           const double *mesh, size_t mesh_len, float *out, int out_on_device);

/* First-pass audit of the fused kernel (GFW_OPT_KERNEL_VARIANT = 3): counters8 = {certified pixels,
 * certified-but-different-from-exact (must stay 0), queued to the exact path, queue overflows,
 * max |approximate - exact| coordinate over certified pixels as f32 bits, addresses outside their buffer (audit mode range-checks),
 * the certificate half-width E of the last frame as f32 bits, 1 spare}.  Call with reset = 1 before
 * the frames to be audited. */
int   gfw_get_audit(gfw_ctx *ctx, unsigned long long *counters8, int reset);
/* (test hooks — device-side math probes, the host-side build check of the run-time specialisation path — are declared in
 * include/gfwarp_testing.h: they are not part of the operator surface) */

/* Static tables the reference exposes through PixelType (pixel_formats.rs):
 * bytes per pixel, element count, default_max_value (0 => None). */
int   gfw_pixel_type_info(int pixel_type, int *bytes_per_pixel,
                          int *element_count, float *default_max_value);

#ifdef __cplusplus
}
#endif
#endif /* GFWARP_H */
