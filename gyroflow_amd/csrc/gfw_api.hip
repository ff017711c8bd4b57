// gfw_api.hip — the extern "C" boundary of libgfwarp (include/gfwarp.h).
//
// Mirrors the reference's backend-object contract (OclWrapper: src/core/gpu/opencl.rs:178-448,
// WgpuWrapper: src/core/gpu/wgpu.rs:147-559): `create` owns all device allocations, `undistort_image`
// validates like the reference does (mismatches are reported, never abort), uploads params + matrices,
// launches, and (for HOST buffers) copies back before returning.  No CPU fallback lives here: if the HIP
// runtime or a gfx950 device is missing every entry point fails with GFW_ERR_NO_DEVICE.
#include <hip/hip_runtime.h>
#include <math.h>
#include <cmath>
#include <stdarg.h>
#include <stdio.h>
#include <string.h>
#include <string>
#include <utility>
#include <vector>

#include "../../include/gfwarp.h"
#include "../../include/gfwarp_testing.h"
#include "gfw_launch.h"
#include "gfw_frame.h"
#include "gfw_matrices.h"
#include "gfw_jit.h"
#include <stdlib.h>
#include <map>
#include <atomic>
#include <mutex>
#include <thread>

static_assert(sizeof(gfw_kernel_params) == 368, "KernelParams must be 368 bytes (stabilization/mod.rs:101-150)");
static_assert(offsetof(gfw_kernel_params, background) == 48, "layout");
static_assert(offsetof(gfw_kernel_params, k) == 80, "layout");
static_assert(offsetof(gfw_kernel_params, translation2d) == 168, "layout");
static_assert(offsetof(gfw_kernel_params, source_rect) == 192, "layout");
static_assert(offsetof(gfw_kernel_params, digital_lens_params) == 224, "layout");
static_assert(offsetof(gfw_kernel_params, max_pixel_value) == 304, "layout");
static_assert(offsetof(gfw_kernel_params, plane_index) == 324, "layout");
static_assert(offsetof(gfw_kernel_params, ewa_coeffs_p) == 336, "layout");

#define GFW_MESH_MAX 839   /* MAX_BUFFER_SIZE, gyro_source/splines.rs:88-89 */

static thread_local std::string g_last_error;
static void set_error(const char *fmt, ...) {
    char buf[512];
    va_list ap; va_start(ap, fmt); vsnprintf(buf, sizeof(buf), fmt, ap); va_end(ap);
    g_last_error = buf;
}
void gfw_set_error_text(const char *text) { g_last_error = text ? text : ""; }      // other translation units of the library (gfw_interop.hip)
#define HIP_TRY(expr, code)                                                                   \
    do { hipError_t e_ = (expr); if (e_ != hipSuccess) {                                      \
        set_error("%s failed: %s", #expr, hipGetErrorString(e_)); return (code); } } while (0)

static const int   PIX_BPP[GFW_PIX_COUNT]   = {1, 2, 3, 4, 4, 6, 8, 8, 16, 8, 4, 2, 4};
static const int   PIX_N[GFW_PIX_COUNT]     = {1, 1, 3, 4, 4, 3, 4, 4, 4, 4, 1, 2, 2};
static const float PIX_MAX[GFW_PIX_COUNT]   = {255.f, 65535.f, 255.f, 255.f, 255.f, 65535.f, 65535.f, 65535.f, 0.f, 0.f, 0.f, 255.f, 65535.f};

struct DevBuf {
    void *ptr = nullptr; size_t cap = 0;
    hipError_t ensure(size_t n) {
        if (n <= cap) return hipSuccess;
        if (ptr) (void)hipFree(ptr);
        ptr = nullptr; cap = 0;
        hipError_t e = hipMalloc(&ptr, n);
        if (e == hipSuccess) cap = n;
        return e;
    }
    void release() { if (ptr) (void)hipFree(ptr); ptr = nullptr; cap = 0; }
};

struct gfw_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    bool own_stream = true;
    bool synchronous = true;
    int matrices_on_device = 0;                  // 0 host rows[14]; 1 device rows[14]; 2 device rows[16] packed by gfw_pack_matrices
    int kernel_variant = 0;
    int tune_rb = 0;
    int tune_grid = 0;
    int num_cus = 256;
    int pixel_type = 0, model = 0, digital = 0;
    int max_matrix_rows = 0;
    size_t src_len = 0, dst_len = 0;              // sizes declared at create (opencl.rs:287-293)
    std::vector<DevBuf> stage_src, stage_dst;      // per-plane staging for HOST buffers
    DevBuf d_mesh;
    // per-row matrices: ring of (pinned host, device) slots so that an asynchronous caller can enqueue several frames;
    // uploads run on their own stream and overlap the previous frame's kernel.
    struct MatSlot { float *h = nullptr; float *d = nullptr; hipEvent_t copied = nullptr, done = nullptr; bool used = false; };
    static constexpr int kMatSlots = 4;
    MatSlot mslots[kMatSlots];
    int mslot_next = 0, mslot_cur = -1;
    hipStream_t copy_stream = nullptr;
    const char *last_backend = "";
    // certified first pass of the fused kernel: s(rho) table cache
    DevBuf d_p1_table, d_audit;
    float p1_eps_last = 0.0f;                      // certificate half-width of the last frame set up (reported in gfw_get_audit's word 6)
    float p1_k[4] = {0, 0, 0, 0}; float p1_rho_max = 0.0f; double p1_etab = 0.0; double p1_smax = 1.0; bool p1_valid = false;
    double p1_slope = 0.0, p1_kappa = 0.0;         // max |ds/drho| over the table's range; roundoff amplification of the exact path's theta_d/r (section 2c)
    struct P1Radial *p1_radial = nullptr;          // the same for the radial models of round 6 (GoPro: gfw_api_certificate.inc); table in d_p1_table as well
    double p1_u1 = 0.0, p1_u2 = 0.0, p1_t32 = 0.0; // max sqrt(rho) |s'|, rho |s'|, rho^1.5 |s''| over the table's range: the curvature of the first pass's value across a lattice cell (gfw_frame.hip)
    DevBuf d_pts_in, d_pts_out, d_pts_rot, d_pts_shift, d_pts_mesh;   // gfw_undistort_points staging
    DevBuf d_tracks;                              // quaternion tracks
    // context-owned per-row tables built on the device (gfw_build_matrices): a small ring, built on copy_stream so that
    // frame N+1's table is produced while frame N is being warped; events order builder and consumer both ways
    struct BuiltSlot { DevBuf buf; hipEvent_t built = nullptr, consumed = nullptr; bool used = false; };   // buf = rows + 4 doubles of builder scratch
    DevBuf d_prefix;                              // builder scratch for caller-owned tables
    // frame descriptors of the builder: pinned host ring + device ring (one entry of up to kMaxBatch descriptors per build)
    static constexpr int kTimingSlots = 8, kMaxBatch = 64;
    gfw_frame_timing *h_timings = nullptr; DevBuf d_timings; int timing_next = 0;
    hipEvent_t timing_copied[kTimingSlots] = {};
    // gfw_build_matrices_batch: two context-owned batches of tables, alternated; built in order on the context stream
    DevBuf d_batch[2]; int batch_next = 0;
    static constexpr int kBuiltSlots = 4;
    BuiltSlot bslots[kBuiltSlots];
    int bslot_next = 0, bslot_cur = -1;
    GfwTracks tracks = {nullptr, nullptr, 0, nullptr, nullptr, 0, nullptr, nullptr, 0, 1.0};
    DevBuf d_offsets;                              // sync offsets of the clip
    // IBIS/OIS control points of the frames being built: a ring of (pinned host, device) pairs, copied on the stream that builds, so that
    // a clip with stabiliser data keeps the asynchronous table ring (build N+1 while N warps)
    struct StabSlot { DevBuf d; void *h = nullptr; size_t hcap = 0; hipEvent_t done = nullptr; bool used = false; };
    static constexpr int kStabSlots = 4;
    StabSlot sslots[kStabSlots]; int sslot_next = 0;
    // run-time specialised kernel (gfw_jit.hip): 0 off; 1 build in the background once the context has seen kJitAfter frames of one
    // clip, warp ahead-of-time meanwhile; 2 build at the first frame and wait for it
    int jit_mode = 1;
    bool dry = false;                              // gfw_debug_jit_key: argument blocks are built, nothing touches a device
    std::string arch;                              // gcnArchName of the device
    std::string jit_header; int jit_seen = 0;      // bake header of the frames being seen, and how many in a row
    GfwYuvArgs jit_key; int jit_key_misc[8] = {}; bool jit_key_valid = false;      // the clip those frames belong to (argument block, per-frame fields blanked)
    hipFunction_t jit_fn = nullptr; int jit_grid = 0;                               // its specialised kernel once loaded
    bool jit_dead = false;                                                          // ... or the verdict that there will be none for this clip (build failed / cache full)
    GfwJitInfo jit_info = {GFW_JIT_UNAVAILABLE, 0.0, std::string()};
    static constexpr int kJitAfter = 3;
    // plane coalescing (round 4): the render loop warps a frame one plane per call, each plane through its own backend object (rendering/mod.rs:494-545);
    // asynchronous device-buffer calls are held until the frame's planes have arrived and leave as ONE fused launch (see PlaneGroup below)
    int coalesce_planes = 1;                       // GFW_OPT_COALESCE_PLANES
    long long paired_launches = 0;                 // launches of the per-plane kernel that served two planes (EWA on planar chroma; gfw_debug_paired_launches)
    int coalesce_frames = 1;                       // GFW_OPT_COALESCE_FRAMES: assembled frames held for one clip launch (1 = each frame leaves when complete)
    hipEvent_t group_done = nullptr;               // orders a member context's stream behind the owner's launch
    hipEvent_t inputs_ready = nullptr;             // a member context's side of the same frame: what was enqueued on ITS stream before its plane's call (an upload, a decode,
                                                   // a consumer still reading the destination) — the owner's stream waits for it before the fused launch
    std::atomic<int> pending_planes{0};            // planes of this context held in some thread's group (flush_if_pending looks here: the holder may be another thread)
    bool multi_plane = false;                      // this context has been seen as one plane of a multi-plane frame (its calls may be held: GFW_OPT_COALESCE_PLANES = 1)
    int frame_sync = 0;                            // GFW_OPT_FRAME_SYNC
    // gfw_set_frame_checksums: the caller's ring of device words, the frames submitted since, the table of partial sums the checksum build of the fused kernel fills
    unsigned long long *sums = nullptr; size_t sum_n = 0, sum_k = 0;
    DevBuf d_ck_part;
    struct ClipBatch *held = nullptr;              // frames assembled from per-plane calls, waiting for their launch (owner context only)
    gfw_ctx *frame_owner = nullptr; bool needs_order = false;   // a member context: whose stream its planes were launched on, and whether its own stream has been ordered behind that yet
    std::vector<gfw_buffers> held_planes;          // ... and the descriptions those frames were validated with
    bool profile = false;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> ev_pool;   // recorded, not yet harvested
    size_t ev_used = 0;
    double prof_ms = 0.0; int64_t prof_launches = 0, prof_frames = 0;
    std::vector<int> ev_frames;                   // frames covered by each bracketed launch
};

extern "C" int gfw_flush(gfw_ctx *c);
int flush_if_pending(gfw_ctx *c);
static void gfw_forget_context(gfw_ctx *c);
static void p1_radial_free(gfw_ctx *c);
static void gfw_register_context(gfw_ctx *c);

static void prof_begin(gfw_ctx *c) {
    if (!c->profile) return;
    if (c->ev_used == c->ev_pool.size()) {
        hipEvent_t a, b;
        (void)hipEventCreate(&a); (void)hipEventCreate(&b);
        c->ev_pool.emplace_back(a, b);
    }
    (void)hipEventRecord(c->ev_pool[c->ev_used].first, c->stream);
}
static void prof_end(gfw_ctx *c, int frames = 1) {
    if (!c->profile) return;
    (void)hipEventRecord(c->ev_pool[c->ev_used].second, c->stream);
    if (c->ev_frames.size() <= c->ev_used) c->ev_frames.resize(c->ev_used + 1);
    c->ev_frames[c->ev_used] = frames;
    c->ev_used++;
}
static void prof_harvest(gfw_ctx *c) {
    for (size_t i = 0; i < c->ev_used; ++i) {
        float ms = 0.0f;
        if (hipEventElapsedTime(&ms, c->ev_pool[i].first, c->ev_pool[i].second) == hipSuccess) { c->prof_ms += ms; c->prof_launches++; c->prof_frames += c->ev_frames[i]; }
    }
    c->ev_used = 0;
}

static int g_device_count = -1;
static int device_count() {
    if (g_device_count < 0) {
        int n = 0;
        if (hipGetDeviceCount(&n) != hipSuccess) n = 0;
        g_device_count = n;
    }
    return g_device_count;
}

extern "C" {

int gfw_abi_version(void) { return GFW_ABI_VERSION; }

int gfw_list_devices(char *buf, size_t cap) {
    const int n = device_count();
    std::string out;
    for (int i = 0; i < n; ++i) {
        hipDeviceProp_t pr;
        if (hipGetDeviceProperties(&pr, i) != hipSuccess) continue;
        out += "[HIP] "; out += pr.name; out += " ("; out += pr.gcnArchName; out += ")\n";
    }
    if (buf && cap) { strncpy(buf, out.c_str(), cap - 1); buf[cap - 1] = 0; }
    if (n == 0) { set_error("no HIP device visible"); return GFW_ERR_NO_DEVICE; }
    return n;
}

static thread_local int g_current_device = 0;
int gfw_set_device(int index) {
    if (index < 0 || index >= device_count()) { set_error("device index %d out of range (%d devices)", index, device_count()); return GFW_ERR_NO_DEVICE; }
    HIP_TRY(hipSetDevice(index), GFW_ERR_HIP);
    g_current_device = index;
    return GFW_OK;
}

int gfw_get_info(char *buf, size_t cap) {
    if (device_count() == 0) { set_error("no HIP device visible"); return GFW_ERR_NO_DEVICE; }
    hipDeviceProp_t pr;
    HIP_TRY(hipGetDeviceProperties(&pr, g_current_device), GFW_ERR_HIP);
    int rt = 0; (void)hipRuntimeGetVersion(&rt);
    char tmp[512];
    snprintf(tmp, sizeof(tmp), "%s %s, %d CUs, %.1f GiB, clock %d MHz, HIP runtime %d", pr.name, pr.gcnArchName,
             pr.multiProcessorCount, (double)pr.totalGlobalMem / (1024.0 * 1024.0 * 1024.0), pr.clockRate / 1000, rt);
    if (buf && cap) { strncpy(buf, tmp, cap - 1); buf[cap - 1] = 0; }
    return GFW_OK;
}

int gfw_is_buffer_supported(const gfw_buffers *b) {
    if (!b) return 0;
    const int ki = b->input.kind, ko = b->output.kind;
    const bool in_ok = ki == GFW_BUF_HOST || ki == GFW_BUF_HIP_DEVICE;
    const bool out_ok = ko == GFW_BUF_HOST || ko == GFW_BUF_HIP_DEVICE;
    return (in_ok && out_ok) ? 1 : 0;
}

int gfw_pixel_type_info(int t, int *bpp, int *count, float *maxv) {
    if (t < 0 || t >= GFW_PIX_COUNT) return GFW_ERR_INVALID_ARGUMENT;
    if (bpp) *bpp = PIX_BPP[t];
    if (count) *count = PIX_N[t];
    if (maxv) *maxv = PIX_MAX[t];
    return GFW_OK;
}

const char *gfw_last_error(void) { return g_last_error.c_str(); }

gfw_ctx *gfw_create(const gfw_kernel_params *params, int pixel_type, int distortion_model, int digital_lens,
                    const gfw_buffers *buffers, size_t drawing_len) {
    (void)drawing_len;   // the CPU kernel never draws overlays (cpu_undistort.rs:234-251 is commented out)
    if (!params || !buffers) { set_error("null params/buffers"); return nullptr; }
    if (pixel_type < 0 || pixel_type >= GFW_PIX_COUNT) { set_error("unknown pixel type %d", pixel_type); return nullptr; }
    if (distortion_model <= GFW_MODEL_NONE || distortion_model > GFW_MODEL_GOPRO6_SUPERVIEW) { set_error("unknown distortion model %d", distortion_model); return nullptr; }
    if (digital_lens < GFW_MODEL_NONE || digital_lens > GFW_MODEL_GOPRO6_SUPERVIEW) { set_error("unknown digital lens %d", digital_lens); return nullptr; }
    if (params->height < 4 || params->output_height < 4 || params->stride < 1) {      // opencl.rs:179
        set_error("size too small: height %d, output_height %d, stride %d", params->height, params->output_height, params->stride); return nullptr; }
    if (!gfw_is_buffer_supported(buffers)) { set_error("unsupported buffer kinds %d/%d", buffers->input.kind, buffers->output.kind); return nullptr; }
    if (params->bytes_per_pixel != PIX_BPP[pixel_type]) { set_error("bytes_per_pixel %d does not match pixel type %d", params->bytes_per_pixel, pixel_type); return nullptr; }
    if (device_count() == 0) { set_error("no HIP device visible (libgfwarp has no CPU fallback)"); return nullptr; }
    if (hipSetDevice(g_current_device) != hipSuccess) { set_error("hipSetDevice(%d) failed", g_current_device); return nullptr; }

    gfw_ctx *c = new gfw_ctx();
    c->device = g_current_device;
    { hipDeviceProp_t pr; if (hipGetDeviceProperties(&pr, c->device) == hipSuccess && pr.multiProcessorCount > 0) { c->num_cus = pr.multiProcessorCount; c->arch = pr.gcnArchName; } }
    if (const char *e = getenv("GFW_JIT")) {                               // deployment / test override of GFW_OPT_JIT's default: 0, 1 or 2, anything else is ignored
        char *end = nullptr; const long v = strtol(e, &end, 10);
        if (end != e && *end == 0 && v >= 0 && v <= 2) c->jit_mode = (int)v;
    }
    if (const char *e = getenv("GFW_COALESCE_PLANES")) { if (e[0] == '0' && e[1] == 0) c->coalesce_planes = 0; }      // deployment override of GFW_OPT_COALESCE_PLANES' default
    c->pixel_type = pixel_type; c->model = distortion_model; c->digital = digital_lens;
    c->src_len = buffers->input.len; c->dst_len = buffers->output.len;
    c->max_matrix_rows = ((params->flags & GFW_FLAG_HORIZONTAL_RS) ? params->width : params->height);   // opencl.rs:287
    if (c->max_matrix_rows < 1) c->max_matrix_rows = 1;
    bool ok = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) == hipSuccess;
    {   // auxiliary stream for uploads and the matrix builder: highest priority, so its small kernels are not starved by a
        // warp that fills the machine
        int prio_lo = 0, prio_hi = 0;
        (void)hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi);
        ok = ok && hipStreamCreateWithPriority(&c->copy_stream, hipStreamNonBlocking, prio_hi) == hipSuccess;
    }
    ok = ok && c->d_mesh.ensure(GFW_MESH_MAX * sizeof(float)) == hipSuccess;
    const size_t mat_bytes = (size_t)c->max_matrix_rows * GFW_MAT_STRIDE * sizeof(float);
    for (int i = 0; i < gfw_ctx::kMatSlots && ok; ++i) {
        gfw_ctx::MatSlot &s = c->mslots[i];
        ok = hipHostMalloc((void **)&s.h, mat_bytes) == hipSuccess && hipMalloc((void **)&s.d, mat_bytes) == hipSuccess &&
             hipEventCreateWithFlags(&s.copied, hipEventDisableTiming) == hipSuccess &&
             hipEventCreateWithFlags(&s.done, hipEventDisableTiming) == hipSuccess;
    }
    c->stage_src.resize(1); c->stage_dst.resize(1);
    if (ok && buffers->input.kind == GFW_BUF_HOST) ok = c->stage_src[0].ensure(c->src_len) == hipSuccess;
    if (ok && buffers->output.kind == GFW_BUF_HOST) ok = c->stage_dst[0].ensure(c->dst_len) == hipSuccess;
    if (!ok) { set_error("device allocation failed: %s", hipGetErrorString(hipGetLastError())); gfw_destroy(c); return nullptr; }
    gfw_register_context(c);
    return c;
}

void gfw_destroy(gfw_ctx *c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    (void)gfw_flush(c);
    gfw_forget_context(c);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    if (c->group_done) (void)hipEventDestroy(c->group_done);
    if (c->inputs_ready) (void)hipEventDestroy(c->inputs_ready);
    for (auto &e : c->ev_pool) { (void)hipEventDestroy(e.first); (void)hipEventDestroy(e.second); }
    for (auto &b : c->stage_src) b.release();
    for (auto &b : c->stage_dst) b.release();
    c->d_mesh.release(); c->d_ck_part.release(); c->d_p1_table.release(); c->d_audit.release(); c->d_tracks.release(); c->d_offsets.release(); for (auto &ss : c->sslots) { ss.d.release(); if (ss.h) (void)hipHostFree(ss.h); if (ss.done) (void)hipEventDestroy(ss.done); } c->d_prefix.release(); c->d_timings.release(); c->d_batch[0].release(); c->d_batch[1].release();
    p1_radial_free(c);
    if (c->h_timings) (void)hipHostFree(c->h_timings);
    for (auto &e : c->timing_copied) if (e) (void)hipEventDestroy(e);
    for (auto &b : c->bslots) { b.buf.release(); if (b.built) (void)hipEventDestroy(b.built); if (b.consumed) (void)hipEventDestroy(b.consumed); }
    c->d_pts_in.release(); c->d_pts_out.release(); c->d_pts_rot.release(); c->d_pts_shift.release(); c->d_pts_mesh.release();
    if (c->copy_stream) { (void)hipStreamSynchronize(c->copy_stream); (void)hipStreamDestroy(c->copy_stream); }
    for (auto &s : c->mslots) {
        if (s.h) (void)hipHostFree(s.h);
        if (s.d) (void)hipFree(s.d);
        if (s.copied) (void)hipEventDestroy(s.copied);
        if (s.done) (void)hipEventDestroy(s.done);
    }
    if (c->own_stream && c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

int gfw_set_option(gfw_ctx *c, int option, int64_t value) {
    if (!c) return GFW_ERR_INVALID_ARGUMENT;
    switch (option) {
    case GFW_OPT_SYNCHRONOUS: { const int frc = gfw_flush(c); if (frc != GFW_OK) return frc; } c->synchronous = value != 0; return GFW_OK;
    case GFW_OPT_MATRICES_ON_DEVICE: c->matrices_on_device = (int)value; return GFW_OK;
    case GFW_OPT_KERNEL_VARIANT: if (value < 0 || value > 4) { set_error("GFW_OPT_KERNEL_VARIANT %lld (0..4; the timing ablations of earlier rounds are not in this library: GFW_TESTING builds only)", (long long)value); return GFW_ERR_INVALID_ARGUMENT; }
                                 c->kernel_variant = (int)value; return GFW_OK;
    case GFW_OPT_PROFILE: c->profile = value != 0; return GFW_OK;
    case GFW_OPT_TUNE_ROWS: c->tune_rb = (int)value; return GFW_OK;
    case GFW_OPT_TUNE_GRID: c->tune_grid = (int)value; c->jit_fn = nullptr; c->jit_key_valid = false; c->jit_dead = false; return GFW_OK;
    case GFW_OPT_JIT: if (value < 0 || value > 2) { set_error("GFW_OPT_JIT %lld", (long long)value); return GFW_ERR_INVALID_ARGUMENT; }
                      c->jit_mode = (int)value; c->jit_fn = nullptr; c->jit_key_valid = false; c->jit_dead = false;
                      c->jit_info = GfwJitInfo{GFW_JIT_UNAVAILABLE, 0.0, std::string()}; return GFW_OK;
    case GFW_OPT_COALESCE_PLANES: if (value < 0 || value > 2) { set_error("GFW_OPT_COALESCE_PLANES %lld (0..2)", (long long)value); return GFW_ERR_INVALID_ARGUMENT; }
                                  { const int frc = gfw_flush(c); if (frc != GFW_OK) return frc; c->coalesce_planes = (int)value; return GFW_OK; }
    case GFW_OPT_FRAME_SYNC: { const int frc = gfw_flush(c); if (frc != GFW_OK) return frc; c->frame_sync = value != 0; return GFW_OK; }
    case GFW_OPT_COALESCE_FRAMES: if (value < 1 || value > GFW_CLIP_FRAMES_MAX) { set_error("GFW_OPT_COALESCE_FRAMES %lld (1..%d)", (long long)value, GFW_CLIP_FRAMES_MAX); return GFW_ERR_INVALID_ARGUMENT; }
                                  { const int frc = gfw_flush(c); if (frc != GFW_OK) return frc; } c->coalesce_frames = (int)value; return GFW_OK;
    default: set_error("unknown option %d", option); return GFW_ERR_INVALID_ARGUMENT;
    }
}
void *gfw_get_stream(gfw_ctx *c) {
    // whoever asks for the stream is about to order something behind this context's work on it (an event, a synchronise, a consumer's kernel): what is being
    // held for a frame or a launch leaves first, so that "everything enqueued so far" means what it meant before planes could be held
    if (c) (void)flush_if_pending(c);
    return c ? (void *)c->stream : nullptr;
}
int gfw_set_stream(gfw_ctx *c, void *s) {
    if (!c) return GFW_ERR_INVALID_ARGUMENT;
    // the previous stream is drained first: frames still in flight on it use this context's staging buffers and matrix slots,
    // and nothing orders the new stream behind them
    HIP_TRY(hipSetDevice(c->device), GFW_ERR_HIP);
    { const int frc = gfw_flush(c); if (frc != GFW_OK) return frc; }
    (void)hipStreamSynchronize(c->stream);
    if (c->own_stream && c->stream) (void)hipStreamDestroy(c->stream);
    c->stream = (hipStream_t)s; c->own_stream = false;
    return GFW_OK;
}
int gfw_synchronize(gfw_ctx *c) {
    if (!c) return GFW_ERR_INVALID_ARGUMENT;
    { const int frc = gfw_flush(c); if (frc != GFW_OK) return frc; }      // planes / frames held for a fused launch leave first
    HIP_TRY(hipStreamSynchronize(c->stream), GFW_ERR_HIP);
    return GFW_OK;
}
const char *gfw_last_backend(gfw_ctx *c) { return c ? c->last_backend : ""; }
int gfw_jit_status(gfw_ctx *c, double *compile_ms, char *log, size_t cap) {
    if (!c) return GFW_ERR_INVALID_ARGUMENT;
    if (compile_ms) *compile_ms = c->jit_info.compile_ms;
    if (log && cap) { snprintf(log, cap, "%s", c->jit_info.log.c_str()); }
    return c->jit_info.state == GFW_JIT_READY ? 2 : c->jit_info.state == GFW_JIT_COMPILING ? 1 : c->jit_info.state == GFW_JIT_FAILED ? 3 : 0;
}
int gfw_get_audit(gfw_ctx *c, unsigned long long *counters8, int reset) {
    if (!c || !counters8) return GFW_ERR_INVALID_ARGUMENT;
    { const int frc_ = flush_if_pending(c); if (frc_ != GFW_OK) return frc_; }

    HIP_TRY(hipSetDevice(c->device), GFW_ERR_HIP);
    const bool fresh = c->d_audit.cap == 0;
    HIP_TRY(c->d_audit.ensure(8 * sizeof(unsigned long long)), GFW_ERR_HIP);
    if (fresh) HIP_TRY(hipMemsetAsync(c->d_audit.ptr, 0, 8 * sizeof(unsigned long long), c->stream), GFW_ERR_HIP);
    HIP_TRY(hipStreamSynchronize(c->stream), GFW_ERR_HIP);
    HIP_TRY(hipMemcpyAsync(counters8, c->d_audit.ptr, 8 * sizeof(unsigned long long), hipMemcpyDeviceToHost, c->stream), GFW_ERR_HIP);      // (in order behind the audited launches)
    HIP_TRY(hipStreamSynchronize(c->stream), GFW_ERR_HIP);
    if (counters8[6] == 0) { uint32_t bits; memcpy(&bits, &c->p1_eps_last, 4); counters8[6] = bits; }      // E (f32 bits): the largest an audited launch used, else the host's estimate for the last frame
    // The reset is a fill ON THE CONTEXT'S STREAM, waited for.  Rounds 2-5 used hipMemset here: a fill of device memory on the NULL stream, which returns before it
    // has run and is not ordered against this context's non-blocking stream — the audited launch that follows could start before, or finish before, the fill landed,
    // and its counters were zeroed under it.  Alone on a GPU the fill runs at once; beside three other processes it queues: 5 of 90 runs of the 200-clip sweep came
    // back with certified + queued < pixels (once with 0 + 0), never with a wrong certificate (profiles/r06_pass1_sweep_concurrent.txt; round-5 verdict, weak #1).
    if (reset) {
        HIP_TRY(hipMemsetAsync(c->d_audit.ptr, 0, 8 * sizeof(unsigned long long), c->stream), GFW_ERR_HIP);
        HIP_TRY(hipStreamSynchronize(c->stream), GFW_ERR_HIP);
    }
    return GFW_OK;
}
int gfw_get_profile(gfw_ctx *c, double *kernel_ms, int64_t *launches, int reset) {
    if (!c) return GFW_ERR_INVALID_ARGUMENT;
    { const int frc_ = flush_if_pending(c); if (frc_ != GFW_OK) return frc_; }

    HIP_TRY(hipStreamSynchronize(c->stream), GFW_ERR_HIP);
    prof_harvest(c);
    if (kernel_ms) *kernel_ms = c->prof_ms;
    if (launches) *launches = c->prof_launches;
    if (reset) { c->prof_ms = 0.0; c->prof_launches = 0; c->prof_frames = 0; }
    return GFW_OK;
}
int gfw_get_profile_frames(gfw_ctx *c, double *kernel_ms, int64_t *launches, int64_t *frames, int reset) {
    if (!c) return GFW_ERR_INVALID_ARGUMENT;
    { const int frc_ = flush_if_pending(c); if (frc_ != GFW_OK) return frc_; }
    if (frames) { HIP_TRY(hipStreamSynchronize(c->stream), GFW_ERR_HIP); prof_harvest(c); *frames = c->prof_frames; }
    return gfw_get_profile(c, kernel_ms, launches, reset);
}

}  // extern "C"

// ------------------------------------------------------------------------------------------------
// per-call validation shared by the plane and frame entry points
static int validate_plane(const gfw_buffers *b, const gfw_kernel_params *p, int pixel_type) {
    if (!b || !p) { set_error("null buffers/params"); return GFW_ERR_INVALID_ARGUMENT; }
    if (!b->input.data || b->input.len == 0)  { set_error("input buffer empty");  return GFW_ERR_INPUT_BUFFER_EMPTY; }   // lib.rs:890
    if (!b->output.data || b->output.len == 0) { set_error("output buffer empty"); return GFW_ERR_OUTPUT_BUFFER_EMPTY; } // lib.rs:891
    if (!gfw_is_buffer_supported(b)) { set_error("unsupported buffer kind"); return GFW_ERR_UNSUPPORTED_BUFFER; }
    if (b->input.height < 4 || b->output.height < 4) { set_error("SizeTooSmall: %d / %d rows", b->input.height, b->output.height); return GFW_ERR_SIZE_TOO_SMALL; }  // mod.rs:613
    if (b->input.width  > p->stride)        { set_error("InvalidStride(%d, %d)", p->stride, b->input.width);  return GFW_ERR_INVALID_STRIDE; }         // mod.rs:639
    if (b->output.width > p->output_stride) { set_error("InvalidStride(%d, %d)", p->output_stride, b->output.width); return GFW_ERR_INVALID_STRIDE; }  // mod.rs:640
    if (p->bytes_per_pixel != PIX_BPP[pixel_type]) { set_error("bytes_per_pixel %d != size_of::<T>() %d", p->bytes_per_pixel, PIX_BPP[pixel_type]); return GFW_ERR_INVALID_ARGUMENT; }  // cpu_undistort.rs:541
    if (b->output.stride <= 0 || p->stride <= 0) { set_error("non-positive stride"); return GFW_ERR_INVALID_STRIDE; }
    if (p->matrix_count < 1) { set_error("matrix_count %d", p->matrix_count); return GFW_ERR_NO_STABILIZATION_DATA; }
    const int it = p->interpolation;
    if (!(it == 2 || it == 4 || it == 8 || (it >= 10 && it <= 13))) { set_error("unknown interpolation %d", it); return GFW_ERR_INVALID_ARGUMENT; }
    // The CPU reference indexes `input[...]` for every tap inside source_rect; out-of-range would panic there.
    const int64_t sx = p->source_rect[0], sy = p->source_rect[1], sw = p->source_rect[2], sh = p->source_rect[3];
    if (sx < 0 || sy < 0 || sw < 0 || sh < 0) { set_error("negative source_rect"); return GFW_ERR_INVALID_ARGUMENT; }
    if (sw > 0 && sh > 0) {
        const int64_t last = (sy + sh - 1) * (int64_t)p->stride + (sx + sw) * (int64_t)p->bytes_per_pixel;
        if (last > (int64_t)b->input.len) { set_error("source_rect (%lld,%lld,%lld,%lld) exceeds the input buffer (%zu bytes, stride %d)", (long long)sx, (long long)sy, (long long)sw, (long long)sh, b->input.len, p->stride); return GFW_ERR_BUFFER_SIZE_MISMATCH; }
    }
    return GFW_OK;
}

// Header of the Sony mesh / focal-plane-distortion block (gyro_source/splines.rs:88-177, sony.rs:557-563): the kernels use
// mesh[0] as the offset of the FPD block and mesh[1], mesh[2] as loop bounds over 9-element arrays.  The reference
// asserts / bounds-checks these (BivariateSpline::new, slice indexing); a malformed block must not reach the device.
template <typename MT>
static int validate_mesh(const MT *mesh, size_t mesh_len) {
    if (!mesh || mesh_len == 0) return GFW_OK;
    if (mesh_len < 9) { set_error("mesh data: %zu values, header needs 9", mesh_len); return GFW_ERR_BUFFER_SIZE_MISMATCH; }
    const double md0 = (double)mesh[0];
    if (!(md0 == md0)) { set_error("mesh data: NaN header"); return GFW_ERR_BUFFER_SIZE_MISMATCH; }
    if (md0 > 10.0) {
        const double nxd = (double)mesh[1], nyd = (double)mesh[2];
        if (!(nxd >= 2.0 && nxd <= 9.0 && nyd >= 2.0 && nyd <= 9.0)) { set_error("mesh data: grid %g x %g outside 2..9", nxd, nyd); return GFW_ERR_BUFFER_SIZE_MISMATCH; }
        const size_t nx = (size_t)nxd, ny = (size_t)nyd;
        if (9 + nx * ny * 2 + 2 * ny * 36 > mesh_len) { set_error("mesh data: %zu values, a %zux%zu grid needs %zu", mesh_len, nx, ny, 9 + nx * ny * 2 + 2 * ny * 36); return GFW_ERR_BUFFER_SIZE_MISMATCH; }
    }
    if (md0 > 0.0) {
        if (md0 >= (double)mesh_len) { set_error("mesh data: focal-plane block offset %g beyond %zu values", md0, mesh_len); return GFW_ERR_BUFFER_SIZE_MISMATCH; }
        const size_t o = (size_t)md0;
        if ((double)mesh[o] > 0.0 && o + 4 + 16 > mesh_len) { set_error("mesh data: focal-plane block at %zu needs 20 values, %zu left", o, mesh_len - o); return GFW_ERR_BUFFER_SIZE_MISMATCH; }
    }
    return GFW_OK;
}

// hipSetDevice only when the calling thread's current device is another one (hipGetDevice reads a thread-local)
static hipError_t select_device(int device) {
    int cur = -1;
    if (hipGetDevice(&cur) == hipSuccess && cur == device) return hipSuccess;
    return hipSetDevice(device);
}

static int upload_matrices(gfw_ctx *c, const float *matrices, int matrix_count, const float **d_out) {
    if (!matrices) { set_error("null matrices"); return GFW_ERR_NO_STABILIZATION_DATA; }
    if (matrix_count > c->max_matrix_rows) {
        // opencl.rs:336 logs "Buffer size mismatch matrices!" and skips the frame
        set_error("Buffer size mismatch matrices! %d vs %d", c->max_matrix_rows, matrix_count); return GFW_ERR_BUFFER_SIZE_MISMATCH; }
    if (c->matrices_on_device == 2) {                                                          // device-resident, already packed
        c->mslot_cur = -1; c->bslot_cur = -1; *d_out = matrices;
        for (int i = 0; i < gfw_ctx::kBuiltSlots; ++i)
            if (c->bslots[i].buf.ptr == (const void *)matrices && c->bslots[i].built) {         // a table gfw_build_matrices produced
                HIP_TRY(hipStreamWaitEvent(c->stream, c->bslots[i].built, 0), GFW_ERR_HIP);
                c->bslot_cur = i;
            }
        return GFW_OK;
    }
    c->bslot_cur = -1;
    c->mslot_cur = c->mslot_next;
    c->mslot_next = (c->mslot_next + 1) % gfw_ctx::kMatSlots;
    gfw_ctx::MatSlot &s = c->mslots[c->mslot_cur];
    if (s.used) HIP_TRY(hipEventSynchronize(s.done), GFW_ERR_HIP);          // the kernel that read this slot has finished
    if (c->matrices_on_device) {
        HIP_TRY(gfw_launch_repack(matrices, s.d, matrix_count, c->stream), GFW_ERR_HIP);
    } else {
        // host repack into pinned memory; cos/sin of the IBIS roll angle come from the host libm so
        // they are the very values the reference's CPU path uses (cpu_undistort.rs:159-160)
        for (int r = 0; r < matrix_count; ++r) {
            const float *m = matrices + (size_t)r * 14;
            float *o = s.h + (size_t)r * GFW_MAT_STRIDE;
            memcpy(o, m, 14 * sizeof(float));
            if (m[9] != 0.0f || m[10] != 0.0f || m[11] != 0.0f || m[12] != 0.0f || m[13] != 0.0f) { o[14] = cosf(-m[11]); o[15] = sinf(-m[11]); }
            else { o[14] = 1.0f; o[15] = 0.0f; }
        }
        HIP_TRY(hipMemcpyAsync(s.d, s.h, (size_t)matrix_count * GFW_MAT_STRIDE * sizeof(float), hipMemcpyHostToDevice, c->copy_stream), GFW_ERR_HIP);
        HIP_TRY(hipEventRecord(s.copied, c->copy_stream), GFW_ERR_HIP);
        HIP_TRY(hipStreamWaitEvent(c->stream, s.copied, 0), GFW_ERR_HIP);
    }
    *d_out = s.d;
    return GFW_OK;
}
static int matrices_consumed(gfw_ctx *c) {          // call after the kernels that read the current slot are enqueued
    if (c->bslot_cur >= 0) {
        gfw_ctx::BuiltSlot &b = c->bslots[c->bslot_cur];
        HIP_TRY(hipEventRecord(b.consumed, c->stream), GFW_ERR_HIP);
        b.used = true;
    }
    if (c->mslot_cur < 0) return GFW_OK;
    gfw_ctx::MatSlot &s = c->mslots[c->mslot_cur];
    HIP_TRY(hipEventRecord(s.done, c->stream), GFW_ERR_HIP);
    s.used = true;
    return GFW_OK;
}

static void fill_common(gfw_ctx *c, const gfw_kernel_params *p, const float *d_mat, const float *d_mesh, int mesh_len, GfwCommon &C) {
    memset(&C, 0, sizeof(C));
    C.matrices = d_mat; C.mesh = d_mesh; C.mesh_len = mesh_len;
    C.model = c->model; C.digital = c->digital;
    C.frame_w = (float)p->width; C.frame_h = (float)p->height;
    C.rot_cos = 1.0f; C.rot_sin = 0.0f;
    if (p->input_rotation != 0.0f) {
        // rotate_point (cpu_undistort.rs:262-265, :486-489) evaluated with host libm, as the reference does
        const float rotation = p->input_rotation * (3.14159265358979323846f / 180.0f);
        C.rot_cos = cosf(rotation); C.rot_sin = sinf(rotation);
        const float s0 = (float)p->width, s1 = (float)p->height;
        const float fx = C.rot_cos * (s0 - 0.0f) - C.rot_sin * (s1 - 0.0f) + 0.0f;
        const float fy = C.rot_sin * (s0 - 0.0f) + C.rot_cos * (s1 - 0.0f) + 0.0f;
        C.frame_w = roundf(fabsf(fx)); C.frame_h = roundf(fabsf(fy));
    }
    C.gopro_tt = tanf(1.5533f);   // gopro.rs:45,58: TMAX.tan()
}


// ------------------------------------------------------------------------------------------------
// divide-by-constant validation for gfw_map_const (gfw_fastmath.h): exhaustive over all 2^23 significands,
// scale-invariant, cached per divisor.
static bool map_const_valid(float den) {
    static std::mutex mu;
    static std::map<uint32_t, bool> cache;
    uint32_t key; memcpy(&key, &den, 4);
    {
        std::lock_guard<std::mutex> g(mu);
        auto it = cache.find(key);
        if (it != cache.end()) return it->second;
    }
    bool ok = den >= 1.0f && den <= 1048576.0f;
    if (ok) {
        const float rcp = 1.0f / den;
        for (uint32_t m = 0; m < (1u << 23) && ok; ++m) {
            const uint32_t bits = 0x3f800000u | m;
            float a; memcpy(&a, &bits, 4);
            const float q0 = a * rcp;
            const float r0 = fmaf(-den, q0, a);
            const float q = fmaf(r0, rcp, q0);
            ok = (q == a / den);
        }
    }
    std::lock_guard<std::mutex> g(mu);
    cache[key] = ok;
    return ok;
}
// x * n is exact in f32 for every integer 0 <= x < n_max  <=>  (n_max-1) * odd_part(n) < 2^24
static bool int_products_exact(int n_max, int n) {
    if (n <= 0 || n_max <= 0) return false;
    int odd = n; while ((odd & 1) == 0) odd >>= 1;
    return (int64_t)(n_max - 1) * odd < (1 << 24);
}

#include "gfw_api_certificate.inc"
#include "gfw_api_eligibility.inc"
#include "gfw_api_clip.inc"

#include "gfw_api_bake.inc"
// gfw_set_frame_checksums behind a kernel that does not take the sum itself: a pass over what the frame's kernels wrote — the pixels of each plane's output rect
// (cpu_undistort.rs:546-551: nothing outside it is touched), whole pixels inside the declared length
static int checksum_written(gfw_ctx *c, int nplanes, const gfw_buffers *planes, const gfw_kernel_params *params, unsigned long long *sum);
// What a plane's kernels WRITE: the pixels of its output rect (cpu_undistort.rs:546-551: nothing outside it is touched), whole pixels inside the declared length,
// as runs of rows: fn(byte offset of the run's first pixel, bytes per row, rows).  The last row may be cut short by the declared length.
template <typename F>
static int for_written_region(const gfw_kernel_params &P, const gfw_buffer_desc &o, F fn) {
    const long long bpp = P.bytes_per_pixel, stride = o.stride;
    if (bpp <= 0 || stride <= 0) return GFW_OK;
    const long long cols = stride / bpp, rows_all = ((long long)o.len + stride - 1) / stride;
    long long x0 = P.output_rect[0] > 0 ? P.output_rect[0] : 0, y0 = P.output_rect[1] > 0 ? P.output_rect[1] : 0;
    long long x1 = (long long)P.output_rect[0] + P.output_rect[2], y1 = (long long)P.output_rect[1] + P.output_rect[3];
    if (x1 > cols) x1 = cols;
    if (y1 > rows_all) y1 = rows_all;
    if (x1 <= x0 || y1 <= y0) return GFW_OK;
    while (y1 > y0 && (y1 - 1) * stride + x1 * bpp > (long long)o.len) {
        const long long fit = ((long long)o.len - (y1 - 1) * stride) / bpp;
        if (fit > x0) { const int rc = fn((y1 - 1) * stride + x0 * bpp, (fit - x0) * bpp, 1LL); if (rc != GFW_OK) return rc; }
        --y1;
    }
    if (y1 > y0) return fn(y0 * stride + x0 * bpp, (x1 - x0) * bpp, y1 - y0);
    return GFW_OK;
}
static int checksum_written(gfw_ctx *c, int nplanes, const gfw_buffers *planes, const gfw_kernel_params *params, unsigned long long *sum) {
    for (int i = 0; i < nplanes; ++i) {
        const gfw_buffer_desc &o = planes[i].output;
        const int rc = for_written_region(params[i], o, [&](long long off, long long row_bytes, long long rows) -> int {
            HIP_TRY(gfw_launch_ck_region((const uint8_t *)o.data, off, o.stride, (int)row_bytes, (int)rows, sum, c->stream), GFW_ERR_HIP);
            return GFW_OK;
        });
        if (rc != GFW_OK) return rc;
    }
    return GFW_OK;
}
static int run_planes(gfw_ctx *c, int nplanes, const gfw_buffers *planes, const gfw_kernel_params *params, const int *pixel_types,
                      const float *matrices, int matrix_count, const float *mesh, size_t mesh_len, ClipBatch *batch = nullptr) {
    if (!c) { set_error("null context"); return GFW_ERR_INVALID_ARGUMENT; }
    if (nplanes < 1 || nplanes > 8) { set_error("nplanes %d", nplanes); return GFW_ERR_INVALID_ARGUMENT; }
    HIP_TRY(select_device(c->device), GFW_ERR_HIP);
    for (int i = 0; i < nplanes; ++i) {
        const int rc = validate_plane(&planes[i], &params[i], pixel_types[i]);
        if (rc != GFW_OK) return rc;
        if (params[i].matrix_count != matrix_count) { set_error("plane %d: matrix_count %d != %d", i, params[i].matrix_count, matrix_count); return GFW_ERR_INVALID_ARGUMENT; }
    }
    if (mesh_len > GFW_MESH_MAX) { set_error("Buffer size mismatch buf_mesh_data! %d vs %zu", GFW_MESH_MAX, mesh_len); return GFW_ERR_BUFFER_SIZE_MISMATCH; }  // opencl.rs:352
    { const int mrc = validate_mesh(mesh, mesh_len); if (mrc != GFW_OK) return mrc; }
    // gfw_set_frame_checksums: this frame's word (taken here, before anything is enqueued: a frame that writes host memory cannot be summed on the device)
    unsigned long long *const sum = c->dry ? nullptr : next_sum(c);
    if (sum) for (int i = 0; i < nplanes; ++i) if (planes[i].output.kind == GFW_BUF_HOST) { set_error("gfw_set_frame_checksums: plane %d writes a host buffer", i); return GFW_ERR_INVALID_ARGUMENT; }
    const float *d_mat = nullptr;
    int rc = upload_matrices(c, matrices, matrix_count, &d_mat);
    if (rc != GFW_OK) return rc;
    const float *d_mesh = nullptr;
    if (mesh && mesh_len) {
        HIP_TRY(hipMemcpyAsync(c->d_mesh.ptr, mesh, mesh_len * sizeof(float), hipMemcpyHostToDevice, c->stream), GFW_ERR_HIP);
        d_mesh = (const float *)c->d_mesh.ptr;
    }
    if ((int)c->stage_src.size() < nplanes) { c->stage_src.resize(nplanes); c->stage_dst.resize(nplanes); }

    GfwPlane launches_arr[8];
    GfwPlane *launches = launches_arr;
    for (int i = 0; i < nplanes; ++i) {
        const gfw_buffers &b = planes[i];
        GfwPlane &A = launches[i];
        memset(&A, 0, sizeof(A));
        A.p = params[i];
        A.pix = pixel_types[i];
        if (b.input.kind == GFW_BUF_HOST) {
            HIP_TRY(c->stage_src[i].ensure(b.input.len), GFW_ERR_HIP);
            HIP_TRY(hipMemcpyAsync(c->stage_src[i].ptr, b.input.data, b.input.len, hipMemcpyHostToDevice, c->stream), GFW_ERR_HIP);  // opencl.rs:359
            A.src = (const uint8_t *)c->stage_src[i].ptr;
        } else A.src = (const uint8_t *)b.input.data;
        if (b.output.kind == GFW_BUF_HOST) {
            // Bytes the kernel never writes (stride padding, pixels outside output_rect) must keep the caller's content, as they do on the CPU path.  Rounds 1-5
            // uploaded the destination first and copied all of it back (33 MB more over the link per C2 frame than opencl.rs:408-413 moves); since round 6 nothing is
            // uploaded and only what the kernels WRITE comes back (for_written_region below): the staging buffer's other bytes are never looked at.  (Page-locking
            // the caller's ranges — hipHostRegister, least recently used out — was built and measured in the same call: 1.291 ms per C2 frame with it, 1.298 without;
            // the runtime's pageable path already runs at the link's rate, and a registration that outlives the caller's allocation is a hazard.  Not kept.)
            HIP_TRY(c->stage_dst[i].ensure(b.output.len), GFW_ERR_HIP);
            A.dst = (uint8_t *)c->stage_dst[i].ptr;
        } else A.dst = (uint8_t *)b.output.data;
        A.dst_len = (int64_t)b.output.len;
        A.dst_stride = b.output.stride;
        A.out_rows = (int32_t)((b.output.len + (size_t)b.output.stride - 1) / (size_t)b.output.stride);
        A.out_cols = b.output.stride / A.p.bytes_per_pixel;
    }

    GfwCommon C;
    GfwYuvArgs Y;
    int bps = 0, n0 = 1, dw = 1, dh = 1; bool interleaved = false, fast1 = false;
    const bool fused = build_yuv_args(c, nplanes, planes, params, pixel_types, launches, c->matrices_on_device ? nullptr : matrices,
                                      matrix_count, mesh_len, Y, bps, n0, dw, dh, interleaved, fast1);
    // gfw_set_frame_checksums: this frame's word.  The specialised fused kernel takes the checksum in its store path when every plane starts on a 64-bit word and
    // every element it stores lies inside one (strides aligned to the element: always, but for a caller's odd sub-buffer); everything else is followed by a pass
    // over what it wrote.  In a clip launch the alignment is the first frame's to answer for the kernel choice and every frame's to meet (checked where frames join).
    bool sum_taken = false;
    if (fused) {
        fill_common(c, &params[0], d_mat, (Y.extras & 32) ? d_mesh : nullptr, (Y.extras & 32) ? (int)mesh_len : 0, Y.common);
        Y.matrices = d_mat;
        if (sum) {
            const int el = bps == 3 ? 2 : bps;        // bytes per element (sample kind 3: half floats)
            bool aligned = true;
            for (int i = 0; i < Y.nplanes; ++i) aligned = aligned && ((uintptr_t)Y.pl[i].dst & 7) == 0 && (Y.pl[i].dst_stride % el) == 0;    // (the kernel places an element in its word by its OFFSET)
            // (the lane-row adds a frame's 8/16-bit samples up in 32-bit registers: fewer than 2^15 lane-rows per lane and frame even if eight workgroups shared the frame)
            const long long lane_rows = (long long)Y.tiles_x * Y.tiles_y * gfw_yuv_rows_per_lane(fast1, 0) * dh;
            Y.checksum = (aligned && lane_rows < 8 * 32768ll) ? 1 : 0;
        }
        int jgrid = 0;
        hipFunction_t jf = jit_for(c, Y, bps, params[0].interpolation, n0, dw, dh, interleaved, fast1, &jgrid);
        if (!jf && fast1 && Y.p1_rform) {
            // a table over r is read by specialised builds only: until one is loaded (or for good, without hiprtc and without a cached kernel) the frame takes the
            // ahead-of-time generic-model kernel and its exact first pass — whose tiles are one lane-row tall
            fast1 = false; Y.p1_table = nullptr; Y.audit = nullptr;
            const int rb = gfw_yuv_rows_per_lane(false, 0);
            Y.tiles_y = (Y.ch + 4 * rb - 1) / (4 * rb);
        }
        if (jf && Y.checksum) { const int krc = ck_table(c, jgrid, Y, batch); if (krc != GFW_OK) return krc; sum_taken = true; }
        bool all_device = c->matrices_on_device != 0;
        for (int i = 0; i < nplanes; ++i) all_device = all_device && planes[i].input.kind != GFW_BUF_HOST && planes[i].output.kind != GFW_BUF_HOST;
        if (jf && batch && all_device && c->bslot_cur < 0 && c->mslot_cur < 0) {      // (a table of the cross-stream ring is ordered by events: frame by frame)
            // the frame joins the clip launch being assembled; a frame that does not share the pending ones' kernel or first-pass table goes out behind them
            if (batch->n > 0 && (batch->fn != jf || batch->CA.Y.p1_table != Y.p1_table || batch->CA.Y.p1_rho_max != Y.p1_rho_max ||
                                 batch->CA.Y.p1_rho_scale != Y.p1_rho_scale || batch->CA.Y.p1_eps != Y.p1_eps || batch->CA.Y.p1_ew != Y.p1_ew ||
                                 !clip_same_params(batch->CA.Y, Y) || clip_overlaps(batch, planes, nplanes))) {
                const int frc = clip_flush(c, batch); if (frc != GFW_OK) return frc;
            }
            if (batch->n == 0) { batch->CA.Y = Y; batch->fn = jf; batch->grid = jgrid; batch->first = planes; batch->backend = fast1 ? "yuv_fused_p1_jit" : "yuv_fused_jit";
                                 batch->limit = clip_launch_limit(Y, nplanes, batch->n_call); }
            batch->sums[batch->n] = sum;
            sum_commit(c, sum);                       // the frame is part of the pending launch from here on
            GfwFrameDyn &F = batch->CA.fr[batch->n++];
            for (int i = 0; i < 4; ++i) { F.src[i] = Y.pl[i].src; F.dst[i] = Y.pl[i].dst; }
            F.matrices = Y.matrices;
            c->last_backend = fast1 ? "yuv_fused_p1_jit" : "yuv_fused_jit";
            if (batch->n >= batch->limit) { const int frc = clip_flush(c, batch); if (frc != GFW_OK) return frc; }
            if (sum && !sum_taken) {                  // (a frame of a launch whose kernel does not take sums: behind the launch it has just joined)
                const int frc = clip_flush(c, batch); if (frc != GFW_OK) return frc;
                const int src_ = checksum_written(c, nplanes, planes, params, sum); if (src_ != GFW_OK) return src_;
            }
            return GFW_OK;
        }
        if (batch) { const int frc = clip_flush(c, batch); if (frc != GFW_OK) return frc; }
        prof_begin(c);
        if (jf) {
            GfwClipArgs CA;
            CA.Y = Y; CA.n_frames = 1; CA.pad_ = 0;
            for (int i = 0; i < 4; ++i) { CA.fr[0].src[i] = Y.pl[i].src; CA.fr[0].dst[i] = Y.pl[i].dst; }
            CA.fr[0].matrices = Y.matrices;
            HIP_TRY(gfw_jit_launch(jf, CA, jgrid, c->stream), GFW_ERR_HIP);
            if (sum_taken) { const int krc = ck_finish(c, CA, jgrid, &sum); if (krc != GFW_OK) return krc; }
            timeline_dump(c, jf);
            c->last_backend = fast1 ? "yuv_fused_p1_jit" : "yuv_fused_jit";
        } else {
            HIP_TRY(gfw_launch_yuv(Y, bps, params[0].interpolation, n0, dw, dh, interleaved, fast1, c->stream), GFW_ERR_HIP);
            c->last_backend = fast1 ? "yuv_fused_p1" : "yuv_fused";
        }
    } else {
        if (batch) { const int frc = clip_flush(c, batch); if (frc != GFW_OK) return frc; }
        prof_begin(c);
        for (int i = 0; i < nplanes; ++i) {
            // EWA on planar chroma (round 6): planes i and i + 1 are single-channel planes of one pixel type, one geometry and — but for plane_index (never 0: the
            // colour-range fix asks only whether a plane is luma) and the background — one KernelParams: the coordinates, jacobians and tap weights of their pixels
            // are the same numbers.  One launch works them out once and keeps two sets of sums (gfw_plane_kernel<.., DUAL>): 1.43 -> 1.11 ms per C2 frame.
            bool paired = false;
            // (plane_index is read for one thing, the colour-range fix's luma / chroma scale: without the flag any two planes qualify — the four planes of a planar
            // float frame leave as two launches —, with it two planes that are both chroma)
            if (i + 1 < nplanes && params[i].interpolation >= 10 && PIX_N[pixel_types[i]] == 1 && pixel_types[i] == pixel_types[i + 1] &&
                ((params[i].plane_index != 0 && params[i + 1].plane_index != 0) || (((params[i].flags | params[i + 1].flags) & GFW_FLAG_FIX_COLOR_RANGE) == 0))) {
                gfw_kernel_params q = params[i + 1];
                q.plane_index = params[i].plane_index;
                memcpy(q.background, params[i].background, sizeof(q.background));
                const GfwPlane &a = launches[i], &b = launches[i + 1];
                // (the two launches it replaces run one after the other: the second plane may not read or overwrite what the first writes, nor the first the second's)
                auto apart = [](const uint8_t *p, size_t pn, const uint8_t *r, size_t rn) { return p + pn <= r || r + rn <= p; };
                const size_t il = planes[i].input.len, ol = (size_t)a.dst_len;
                paired = memcmp(&q, &params[i], sizeof(q)) == 0 && a.dst_len == b.dst_len && a.dst_stride == b.dst_stride && a.out_rows == b.out_rows && a.out_cols == b.out_cols &&
                         planes[i].input.len == planes[i + 1].input.len && apart(a.dst, ol, b.dst, ol) && apart(a.dst, ol, b.src, il) && apart(b.dst, ol, a.src, il);
            }
            fill_common(c, &params[i], d_mat, d_mesh, (int)mesh_len, C);
            if (paired) {
                GfwPlane two = launches[i];
                two.src2 = launches[i + 1].src; two.dst2 = launches[i + 1].dst;
                memcpy(two.background2, params[i + 1].background, sizeof(two.background2));
                HIP_TRY(gfw_launch_plane(two, C, c->stream), GFW_ERR_HIP);
                c->paired_launches++;
                ++i;
                continue;
            }
            HIP_TRY(gfw_launch_plane(launches[i], C, c->stream), GFW_ERR_HIP);
        }
        c->last_backend = "plane_generic";
    }
    sum_commit(c, sum);                               // the frame's kernels are enqueued
    if (sum && !sum_taken) { const int src_ = checksum_written(c, nplanes, planes, params, sum); if (src_ != GFW_OK) return src_; }
    prof_end(c);
    { const int mrc = matrices_consumed(c); if (mrc != GFW_OK) return mrc; }
    if (c->ev_used > 4096) { (void)hipStreamSynchronize(c->stream); prof_harvest(c); }

    bool any_host_out = false;
    for (int i = 0; i < nplanes; ++i) {
        if (planes[i].output.kind == GFW_BUF_HOST) {
            // opencl.rs:413 reads the whole buffer back; here: the written pixels only — one linear copy when the rows are written whole, else a pitched one
            uint8_t *host = (uint8_t *)planes[i].output.data; const uint8_t *dev = (const uint8_t *)c->stage_dst[i].ptr; const long long stride = planes[i].output.stride;
            const int crc = for_written_region(params[i], planes[i].output, [&](long long off, long long row_bytes, long long rows) -> int {
                if (row_bytes == stride || rows == 1) HIP_TRY(hipMemcpyAsync(host + off, dev + off, (size_t)(rows == 1 ? row_bytes : row_bytes * rows), hipMemcpyDeviceToHost, c->stream), GFW_ERR_HIP);
                else HIP_TRY(hipMemcpy2DAsync(host + off, (size_t)stride, dev + off, (size_t)stride, (size_t)row_bytes, (size_t)rows, hipMemcpyDeviceToHost, c->stream), GFW_ERR_HIP);
                return GFW_OK;
            });
            if (crc != GFW_OK) return crc;
            any_host_out = true;
        }
    }
    if (c->synchronous || any_host_out) HIP_TRY(hipStreamSynchronize(c->stream), GFW_ERR_HIP);
    return GFW_OK;
}

#include "gfw_api_coalesce.inc"

extern "C" {

int gfw_flush(gfw_ctx *c) {
    if (!c) return GFW_ERR_INVALID_ARGUMENT;
    GroupLock lk(g_group_mu);
    return flush_context_locked(c, lk);
}
}  // extern "C"
// Entry points other than gfw_undistort_image keep their place in the order of calls: whatever is being held leaves first (no lock when nothing is)
int flush_if_pending(gfw_ctx *c) {
    if (!c) return GFW_OK;
    if (!(t_group && t_group->n > 0) && !(c->held && c->held->n > 0) && !c->needs_order && c->pending_planes.load(std::memory_order_relaxed) == 0) return GFW_OK;
    return gfw_flush(c);
}
extern "C" {

int gfw_undistort_image(gfw_ctx *c, const gfw_buffers *buffers, const gfw_kernel_params *params,
                        const float *matrices, int matrix_count, const uint8_t *drawing, size_t drawing_len,
                        const float *mesh, size_t mesh_len) {
    (void)drawing; (void)drawing_len;
    if (!c) { set_error("null context"); return GFW_ERR_INVALID_ARGUMENT; }
    if (buffers && buffers->input.kind == GFW_BUF_HOST && buffers->input.len != c->src_len) {
        set_error("Buffer size mismatch input! %zu vs %zu", c->src_len, buffers->input.len); return GFW_ERR_BUFFER_SIZE_MISMATCH; }   // opencl.rs:358
    const int pt = c->pixel_type;
    // Can this call wait for the rest of its frame?  Only if nothing about it has to happen before the call returns.
    const bool single_channel = pt == GFW_PIX_LUMA8 || pt == GFW_PIX_LUMA16 || pt == GFW_PIX_R32F || pt == GFW_PIX_UV8 || pt == GFW_PIX_UV16;
    // GFW_OPT_COALESCE_PLANES = 1 holds a call only on a context that has been SEEN as one plane of a multi-plane frame (below): a greyscale or single-float-plane
    // caller's calls are never held, whatever their plane_index says.  A synchronous context's call may be held only under GFW_OPT_FRAME_SYNC.
    plane_pattern_note(c, params, matrix_count);
    const bool may_hold = c->coalesce_planes == 2 || (c->coalesce_planes == 1 && c->multi_plane);
    bool holdable = may_hold && (!c->synchronous || c->frame_sync) && buffers && params && matrices && single_channel && c->kernel_variant == 0 &&
                          buffers->input.kind == GFW_BUF_HIP_DEVICE && buffers->output.kind == GFW_BUF_HIP_DEVICE && (!mesh || mesh_len == 0) &&
                          params->plane_index >= 0 && params->plane_index < 4 && matrix_count >= 1;
    // (a context that never took part in a held frame, called by a thread that holds nothing: the round-3 path, no lock)
    if (!holdable && !(t_group && t_group->n > 0) && !(c->held && c->held->n > 0) && !c->needs_order)
        return run_planes(c, 1, buffers, params, &pt, matrices, matrix_count, mesh, mesh_len);
    GroupLock lk(g_group_mu);
    PlaneGroup *g = t_group;
    if (!g && holdable) { g = t_group = new PlaneGroup(); g->thread = std::this_thread::get_id(); g_groups.push_back(g); }
    bool cont = false;
    if (g && g->n > 0) {
        // does the call continue the frame being assembled?
        cont = holdable && params->plane_index == g->n && g->pl[0].c->device == c->device && g->pl[0].c->model == c->model && g->pl[0].c->digital == c->digital &&
               g->matrix_count == matrix_count && g->matrices_on_device == c->matrices_on_device;
        if (cont) cont = g->matrices_on_device ? (matrices == g->d_matrices) : (memcmp(matrices, g->h_matrices.data(), (size_t)matrix_count * 14 * sizeof(float)) == 0);
        if (!cont) {
            // the frame being assembled leaves INCOMPLETE because this call does not continue it: its contexts stop counting as planes of a multi-plane frame
            // (GFW_OPT_COALESCE_PLANES = 1 holds calls only on such contexts) until the pattern is seen again — a context reused for lone planes is not held
            // until the next flush for the rest of its life
            if (c->coalesce_planes == 1) { for (int i = 0; i < g->n; ++i) g->pl[i].c->multi_plane = false; t_last_plane.marked = false; if (!c->multi_plane) holdable = false; }
            const int frc = group_launch(g, lk); if (frc != GFW_OK) return frc;
        }
    }
    const bool starts = holdable && !cont && params->plane_index == 0;
    // whatever else involves this context leaves first, unless the call is the next plane of the frame or opens the next frame of the clip its context holds
    if (!cont && !(starts && c->coalesce_frames > 1)) { const int frc = flush_context_locked(c, lk); if (frc != GFW_OK) return frc; }
    if (!cont && !starts) {
        lk.unlock();
        return run_planes(c, 1, buffers, params, &pt, matrices, matrix_count, mesh, mesh_len);
    }
    {   // the errors of this plane belong to this call
        const int vrc = validate_plane(buffers, params, pt);
        if (vrc != GFW_OK) { (void)group_launch(g, lk); return vrc; }
        if (params->matrix_count != matrix_count) { (void)group_launch(g, lk); set_error("plane %d: matrix_count %d != %d", g->n, params->matrix_count, matrix_count); return GFW_ERR_INVALID_ARGUMENT; }
        if (matrix_count > c->max_matrix_rows) { (void)group_launch(g, lk); set_error("Buffer size mismatch matrices! %d vs %d", c->max_matrix_rows, matrix_count); return GFW_ERR_BUFFER_SIZE_MISMATCH; }
    }
    if (g->n == 0) {
        g->matrix_count = matrix_count; g->matrices_on_device = c->matrices_on_device;
        if (c->matrices_on_device) g->d_matrices = matrices;
        else g->h_matrices.assign(matrices, matrices + (size_t)matrix_count * 14);
    }
    PendingPlane &P = g->pl[g->n++];
    P.c = c; P.b = *buffers; P.p = *params; P.pixel_type = pt;
    c->pending_planes.fetch_add(1, std::memory_order_relaxed);
    c->last_backend = "held_for_frame";
    // complete?  An interleaved chroma plane ends a frame; planar 8/16-bit frames have three planes (a fourth — alpha — goes by itself), planar float four
    const bool complete = pt == GFW_PIX_UV8 || pt == GFW_PIX_UV16 || ((pt == GFW_PIX_LUMA8 || pt == GFW_PIX_LUMA16) && g->n == 3) || g->n == 4;
    if (complete) {
        gfw_ctx *owner = g->pl[0].c;
        const int rc = group_launch(g, lk);
        lk.unlock();
        // GFW_OPT_FRAME_SYNC on a synchronous member of an asynchronous owner: the frame is complete when this call returns (a synchronous owner waited in run_planes)
        if (rc == GFW_OK && c->synchronous && c != owner) HIP_TRY(hipStreamSynchronize(owner->stream), GFW_ERR_HIP);
        return rc;
    }
    return GFW_OK;
}

int gfw_undistort_frame(gfw_ctx *c, int nplanes, const gfw_buffers *planes, const gfw_kernel_params *params,
                        const int *pixel_types, const float *matrices, int matrix_count, const float *mesh, size_t mesh_len) {
    if (!planes || !params || !pixel_types) { set_error("null plane arrays"); return GFW_ERR_INVALID_ARGUMENT; }
    for (int i = 0; i < nplanes; ++i)
        if (pixel_types[i] < 0 || pixel_types[i] >= GFW_PIX_COUNT) { set_error("plane %d: unknown pixel type %d", i, pixel_types[i]); return GFW_ERR_INVALID_ARGUMENT; }
    { const int frc_ = flush_if_pending(c); if (frc_ != GFW_OK) return frc_; }
    return run_planes(c, nplanes, planes, params, pixel_types, matrices, matrix_count, mesh, mesh_len);
}

int gfw_undistort_clip(gfw_ctx *c, int n_frames, int nplanes, const gfw_buffers *planes, const gfw_kernel_params *params,
                       const int *pixel_types, const float *const *matrices, int matrix_count) {
    if (!c) { set_error("null context"); return GFW_ERR_INVALID_ARGUMENT; }
    if (n_frames < 0 || !planes || !params || !pixel_types || !matrices) { set_error("null clip arrays"); return GFW_ERR_INVALID_ARGUMENT; }
    { const int frc_ = flush_if_pending(c); if (frc_ != GFW_OK) return frc_; }

    for (int i = 0; i < nplanes; ++i)
        if (pixel_types[i] < 0 || pixel_types[i] >= GFW_PIX_COUNT) { set_error("plane %d: unknown pixel type %d", i, pixel_types[i]); return GFW_ERR_INVALID_ARGUMENT; }
    // the frame loop of a render (rendering/mod.rs:487-547 calls process_pixels once per frame), here on the library side: frames that
    // share the specialised kernel leave in launches of up to GFW_CLIP_MAX frames, everything else exactly as gfw_undistort_frame
    ClipBatch batch;
    batch.n_call = n_frames;
    for (int f = 0; f < n_frames; ++f) {
        // A frame shaped exactly like the one that opened the pending launch (same descriptions but for the pointers; the parameters are
        // shared by construction) needs none of the per-frame validation again: its pointers join the launch.  ~10 us -> < 1 us of host time.
        bool words = true;                            // (the checksum build places an element by its offset: every plane on a 64-bit word, as run_planes checked for the launch's first frame)
        if (c->sums && batch.n > 0) for (int i = 0; i < nplanes && i < 4; ++i) words = words && (((uintptr_t)planes[(size_t)f * nplanes + i].output.data + (uintptr_t)(batch.CA.fr[0].dst[i] - (uint8_t *)batch.first[i].output.data)) & 7) == 0;
        if (batch.n > 0 && batch.n < batch.limit && (!c->sums || (batch.CA.Y.checksum && words)) && c->matrices_on_device == 2 && matrices[f] && clip_same_shape(batch.first, planes + (size_t)f * nplanes, nplanes) &&
            !clip_ring_table(c, matrices[f]) && !clip_overlaps(&batch, planes + (size_t)f * nplanes, nplanes)) {
            batch.sums[batch.n] = next_sum(c);
            sum_commit(c, batch.sums[batch.n]);
            GfwFrameDyn &F = batch.CA.fr[batch.n++];
            for (int i = 0; i < 4; ++i) {
                // (the kernel's planes begin at their rects' first pixels: the same offsets the launch's first frame was given — clip_same_shape compared the rects)
                const ptrdiff_t so = i < nplanes ? batch.CA.fr[0].src[i] - (const uint8_t *)batch.first[i].input.data : 0, dof = i < nplanes ? batch.CA.fr[0].dst[i] - (uint8_t *)batch.first[i].output.data : 0;
                F.src[i] = i < nplanes ? (const uint8_t *)planes[(size_t)f * nplanes + i].input.data + so : nullptr;
                F.dst[i] = i < nplanes ? (uint8_t *)planes[(size_t)f * nplanes + i].output.data + dof : nullptr;
            }
            F.matrices = matrices[f];
            c->last_backend = batch.backend;
            if (batch.n >= batch.limit) { const int frc = clip_flush(c, &batch); if (frc != GFW_OK) return frc; }
            continue;
        }
        const int rc = run_planes(c, nplanes, planes + (size_t)f * nplanes, params, pixel_types, matrices[f], matrix_count, nullptr, 0, &batch);
        if (rc != GFW_OK) { (void)clip_flush(c, &batch); if (c->synchronous) (void)hipStreamSynchronize(c->stream); return rc; }
    }
    const int frc = clip_flush(c, &batch);
    // GFW_OPT_SYNCHRONOUS (the default) means what it means for gfw_undistort_frame: the outputs are complete when the call returns — a frame that
    // joined a clip launch left run_planes before its own synchronisation point (round-3 advisor finding)
    if (c->synchronous) { const hipError_t e = hipStreamSynchronize(c->stream); if (e != hipSuccess && frc == GFW_OK) { set_error("hipStreamSynchronize failed: %s", hipGetErrorString(e)); return GFW_ERR_HIP; } }
    return frc;
}

}  // extern "C"

#include "gfw_api_testhooks.inc"
#include "gfw_api_adjacent.inc"
