// gfw_kernels.hip — kernel dispatch by PixelType + the small utility kernels of libgfwarp.
#include <hip/hip_runtime.h>
#include "gfw_launch.h"
#include "gfw_fastmath.h"
#include "gfw_frame.h"

#define GFW_DECL(n) hipError_t gfw_launch_plane_pix##n(const GfwPlane &A, const GfwCommon &C, hipStream_t s);
GFW_DECL(0) GFW_DECL(1) GFW_DECL(2) GFW_DECL(3) GFW_DECL(4) GFW_DECL(5) GFW_DECL(6)
GFW_DECL(7) GFW_DECL(8) GFW_DECL(9) GFW_DECL(10) GFW_DECL(11) GFW_DECL(12)

hipError_t gfw_launch_plane(const GfwPlane &A, const GfwCommon &C, hipStream_t s) {
    switch (A.pix) {
#define GFW_CASE(n) case n: return gfw_launch_plane_pix##n(A, C, s);
    GFW_CASE(0) GFW_CASE(1) GFW_CASE(2) GFW_CASE(3) GFW_CASE(4) GFW_CASE(5) GFW_CASE(6)
    GFW_CASE(7) GFW_CASE(8) GFW_CASE(9) GFW_CASE(10) GFW_CASE(11) GFW_CASE(12)
    default: return hipErrorInvalidValue;
    }
}

// ---------------------------------------------------------------------------- fused frame kernel dispatch
#define GFW_DECL_YUV(K, I) hipError_t gfw_launch_yuv_kind##K##_taps##I(const GfwYuvArgs &A, int n0, int dw, int dh, bool interleaved, bool fast1, hipStream_t s);
GFW_DECL_YUV(1, 2) GFW_DECL_YUV(1, 4) GFW_DECL_YUV(1, 8) GFW_DECL_YUV(2, 2) GFW_DECL_YUV(2, 4) GFW_DECL_YUV(2, 8) GFW_DECL_YUV(4, 2) GFW_DECL_YUV(4, 4) GFW_DECL_YUV(4, 8) GFW_DECL_YUV(3, 2) GFW_DECL_YUV(3, 4) GFW_DECL_YUV(3, 8)
int gfw_yuv_rows_per_lane(bool fast1, int tune_rb) {
    (void)tune_rb;
    return fast1 ? GFW_YUV_RB_FAST : GFW_YUV_RB_EXACT;
}
hipError_t gfw_launch_yuv(const GfwYuvArgs &A, int sample_kind, int taps, int n0, int dw, int dh, bool interleaved, bool fast1, hipStream_t s) {
#define GFW_CASE_YUV(K, I) if (sample_kind == K && taps == I) return gfw_launch_yuv_kind##K##_taps##I(A, n0, dw, dh, interleaved, fast1, s);
    GFW_CASE_YUV(1, 2) GFW_CASE_YUV(1, 4) GFW_CASE_YUV(1, 8) GFW_CASE_YUV(2, 2) GFW_CASE_YUV(2, 4) GFW_CASE_YUV(2, 8) GFW_CASE_YUV(4, 2) GFW_CASE_YUV(4, 4) GFW_CASE_YUV(4, 8) GFW_CASE_YUV(3, 2) GFW_CASE_YUV(3, 4) GFW_CASE_YUV(3, 8)
    return hipErrorInvalidValue;
}

// ---------------------------------------------------------------------------- STMap coordinate export
// src/core/stmap.rs:87-109: the "undist" map is the warp's coordinate stage alone — rolling-shutter row pick, then
// rotate_and_distort — written as two f32 per pixel instead of being sampled.
template <int MODEL>
__global__ __launch_bounds__(256) void gfw_stmap_kernel(const gfw_kernel_params P, const GfwCommon C, int width, int height, float *coords) {
    const int x = blockIdx.x * 64 + threadIdx.x, y = blockIdx.y * 4 + threadIdx.y;
    if (x >= width || y >= height) return;
    const float fx = (float)x, fy = (float)y;
    const float r_limit_sq = P.r_limit * P.r_limit;
    const bool hrs = (P.flags & 16) == 16;
    const int lim = hrs ? P.width : P.height;
    int sy = max(min(gfw_f2i(gfw_round(hrs ? fx : fy)), lim), 0);
    if (P.matrix_count > 1) {
        const GfwPt pt = gfw_rotate_and_distort<MODEL>(fx, fy, P.matrix_count / 2, P, C, r_limit_sq);
        if (pt.ok) sy = max(min(gfw_f2i(gfw_round(hrs ? pt.x : pt.y)), lim), 0);
    }
    const GfwPt uv = gfw_rotate_and_distort<MODEL>(fx, fy, min(sy, P.matrix_count - 1), P, C, r_limit_sq);
    if (uv.ok) *reinterpret_cast<float2 *>(coords + ((size_t)y * width + x) * 2) = float2{uv.x, uv.y};
}
hipError_t gfw_launch_stmap(const gfw_kernel_params &P, const GfwCommon &C, int width, int height, float *coords, hipStream_t s) {
    dim3 grid((width + 63) / 64, (height + 3) / 4), block(64, 4);
    if (C.model == GFW_MODEL_OPENCV_FISHEYE && C.mesh_len == 0) hipLaunchKernelGGL(gfw_stmap_kernel<GFW_MODEL_OPENCV_FISHEYE>, grid, block, 0, s, P, C, width, height, coords);
    else hipLaunchKernelGGL(gfw_stmap_kernel<-1>, grid, block, 0, s, P, C, width, height, coords);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------- inverse point map
// `undistort_points` (cpu_undistort.rs:652-858): source-image point -> stabilised output
// coordinate; the STMap "dist" pass (stmap.rs:123-127) runs it over the pixel grid.  One point per lane; the rotation
// (and optional IBIS/OIS shift) row is picked per point / grid row / grid column.  shifts rows are 6 floats:
// sx, sy, cos(angle), sin(angle), ox, oy with the trig evaluated by the host libm (gfw_api.hip).
// undistort_points' lens-correction branch (cpu_undistort.rs:785-851): Newton solve of amount*o + (1-amount)*R(o) = pt,
// R = the render's forward map (digital undistort -> /out_f -> radial undistort -> refraction -> *out_f), :804-826.
struct GfwLc { float out_c0, out_c1, out_f0, out_f1, amount, factor, fov; };
template <int MODEL>
__device__ inline float2 gfw_lc_r_of(const GfwLc &L, float o0, float o1, const gfw_kernel_params &P, const GfwCommon &C) {
    float q0 = o0, q1 = o1;
    if (C.digital != GFW_MODEL_NONE) {
        const float uz0 = (q0 - L.out_c0) * L.fov + L.out_c0, uz1 = (q1 - L.out_c1) * L.fov + L.out_c1;
        const GfwPt d = gfw_lens::digital_undistort(C.digital, uz0, uz1, P);
        if (d.ok) { q0 = (d.x - L.out_c0) / L.fov + L.out_c0; q1 = (d.y - L.out_c1) / L.fov + L.out_c1; }
    }
    float n0 = (q0 - L.out_c0) / L.out_f0, n1 = (q1 - L.out_c1) / L.out_f1;
    const GfwPt d = gfw_lens::undistort<MODEL>(C.model, n0, n1, P, C);
    if (d.ok) { n0 = d.x; n1 = d.y; }
    if (P.light_refraction_coefficient != 1.0f && P.light_refraction_coefficient > 0.0f) {
        const float r = sqrtf(n0 * n0 + n1 * n1);
        if (r != 0.0f) {
            const float sin_theta_d = (r / sqrtf(1.0f + r * r)) / P.light_refraction_coefficient;
            const float r_d = sin_theta_d / sqrtf(1.0f - sin_theta_d * sin_theta_d);
            const float sc = r_d / r;
            n0 = n0 * sc; n1 = n1 * sc;
        }
    }
    return float2{(n0 * L.out_f0) + L.out_c0, (n1 * L.out_f1) + L.out_c1};
}
__device__ __forceinline__ bool gfw_finite(float x) { return fabsf(x) < __builtin_inff(); }   // false for inf and NaN
template <int MODEL>
__device__ inline float2 gfw_lc_solve(const GfwLc &L, float p0, float p1, const gfw_kernel_params &P, const GfwCommon &C) {
    float inv0, inv1;
    {
        const float n0 = (p0 - L.out_c0) / L.out_f0, n1 = (p1 - L.out_c1) / L.out_f1;
        float d0, d1;
        gfw_lens::distort<MODEL>(C.model, n0, n1, 1.0f, P, C, d0, d1);
        inv0 = (d0 * L.out_f0) + L.out_c0; inv1 = (d1 * L.out_f1) + L.out_c1;
        if (C.digital != GFW_MODEL_NONE) {
            const float uz0 = (inv0 - L.out_c0) * L.fov + L.out_c0, uz1 = (inv1 - L.out_c1) * L.fov + L.out_c1;
            float dd0, dd1;
            gfw_lens::digital_distort(C.digital, uz0, uz1, P, dd0, dd1);
            inv0 = (dd0 - L.out_c0) / L.fov + L.out_c0; inv1 = (dd1 - L.out_c1) / L.fov + L.out_c1;
        }
    }
    float o0 = p0, o1 = p1;
    if (gfw_finite(inv0) && gfw_finite(inv1)) { o0 = inv0 * L.factor + p0 * L.amount; o1 = inv1 * L.factor + p1 * L.amount; }
    #pragma unroll 1
    for (int it = 0; it < 10; ++it) {
        const float2 r = gfw_lc_r_of<MODEL>(L, o0, o1, P, C);
        const float g0 = L.amount * o0 + L.factor * r.x - p0, g1 = L.amount * o1 + L.factor * r.y - p1;
        if (fabsf(g0) < 0.02f && fabsf(g1) < 0.02f) break;
        const float eps = 1.0f;
        const float2 rx = gfw_lc_r_of<MODEL>(L, o0 + eps, o1, P, C);
        const float2 ry = gfw_lc_r_of<MODEL>(L, o0, o1 + eps, P, C);
        const float j11 = L.amount + L.factor * (rx.x - r.x) / eps, j21 = L.factor * (rx.y - r.y) / eps;
        const float j12 = L.factor * (ry.x - r.x) / eps,            j22 = L.amount + L.factor * (ry.y - r.y) / eps;
        const float det = j11 * j22 - j12 * j21;
        if (!gfw_finite(det) || fabsf(det) < 1e-9f) break;
        const float dx = ( j22 * g0 - j12 * g1) / det;
        const float dy = (-j21 * g0 + j11 * g1) / det;
        if (!gfw_finite(dx) || !gfw_finite(dy)) break;
        o0 = o0 - dx; o1 = o1 - dy;
    }
    return float2{o0, o1};
}

template <int MODEL>
__global__ __launch_bounds__(256) void gfw_points_kernel(const gfw_kernel_params P, const GfwCommon C, const GfwPointsArgs A) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= A.n) return;
    float x, y; size_t gx = 0, gy = 0;
    if (A.points) { const float2 p = reinterpret_cast<const float2 *>(A.points)[i]; x = p.x; y = p.y; }
    else { gy = i / (size_t)A.grid_w; gx = i - gy * (size_t)A.grid_w; x = (float)gx; y = (float)gy; }
    size_t index = A.index_mode == 1 ? i : A.index_mode == 2 ? gy : A.index_mode == 3 ? gx : 0;
    if (index >= (size_t)A.rotation_count) index = 0;
    if (P.input_horizontal_stretch > 0.001f) x *= P.input_horizontal_stretch;                 // :704-705
    if (P.input_vertical_stretch   > 0.001f) y *= P.input_vertical_stretch;
    if (C.digital != GFW_MODEL_NONE) {                                                         // :707-712
        const GfwPt d = gfw_lens::digital_undistort(C.digital, x, y, P);
        if (d.ok) { x = d.x; y = d.y; }
    }
    if (A.mesh_len > 0) {
        const double *md = A.mesh;
        if (md[0] > 0.0 && md[gfw_mesh::d2us(md[0])] > 0.0) {                                  // :715-738
            const int64_t o = gfw_mesh::d2us(md[0]);
            const double ms1 = md[4];
            const float or0 = (float)md[5], or1 = (float)md[6], cs0 = (float)md[7], cs1 = (float)md[8];
            const double grid = ms1 / 8.0;
            x = gfw_map_coord(x, 0.0f, (float)P.width,  or0, or0 + cs0);
            y = gfw_map_coord(y, 0.0f, (float)P.height, or1, or1 + cs1);
            const int64_t idx = gfw_mesh::d2us(fmin(fmax(floor((double)y / grid), 0.0), 7.0));
            const double delta = (double)y - grid * (double)idx;
            x += (float)(md[o + 4 + idx * 2 + 0] * delta);
            y += (float)(md[o + 4 + idx * 2 + 1] * delta);
            for (int64_t j = 0; j < idx; ++j) {
                x += (float)(md[o + 4 + j * 2 + 0] * grid);
                y += (float)(md[o + 4 + j * 2 + 1] * grid);
            }
            x = gfw_map_coord(x, or0, or0 + cs0, 0.0f, (float)P.width);
            y = gfw_map_coord(y, or1, or1 + cs1, 0.0f, (float)P.height);
        }
        if (md[0] > 10.0) {                                                                    // :740-752
            const double ms0 = md[3], ms1 = md[4];
            const float or0 = (float)md[5], or1 = (float)md[6], cs0 = (float)md[7], cs1 = (float)md[8];
            x = gfw_map_coord(x, 0.0f, (float)P.width,  or0, or0 + cs0);
            y = gfw_map_coord(y, 0.0f, (float)P.height, or1, or1 + cs1);
            const int nx = (int)gfw_mesh::d2us(md[1]), ny = (int)gfw_mesh::d2us(md[2]);
            const double nxp = gfw_mesh::bivariate(nx, ny, ms0, ms1, md, 0, (double)x, (double)y);
            const double nyp = gfw_mesh::bivariate(nx, ny, ms0, ms1, md, 1, (double)x, (double)y);
            x = gfw_map_coord((float)nxp, or0, or0 + cs0, 0.0f, (float)P.width);
            y = gfw_map_coord((float)nyp, or1, or1 + cs1, 0.0f, (float)P.height);
        }
    }
    const float c0 = P.c[0], c1 = P.c[1];
    if (A.shifts) {                                                                            // :754-763
        const float *s = A.shifts + index * 6;
        x = x - c0 - s[4] + s[0];
        y = y - c1 - s[5] + s[1];
        x = s[2] * x - s[3] * y + c0;
        y = s[3] * x + s[2] * y + c1;                        // the reference rotates y with the already-rotated x
    }
    const float pwx = (x - c0) / P.f[0], pwy = (y - c1) / P.f[1];                              // :765
    const GfwPt pt = gfw_lens::undistort<MODEL>(C.model, pwx, pwy, P, C);
    float2 o = float2{-1000000.0f, -1000000.0f};                                               // :855
    if (pt.ok) {
        float ptx = pt.x, pty = pt.y;
        if (P.light_refraction_coefficient != 1.0f && P.light_refraction_coefficient > 0.0f) { // :770-779
            const float rr = sqrtf(ptx * ptx + pty * pty);
            if (rr != 0.0f) {
                const float sin_theta_d = (rr / sqrtf(1.0f + rr * rr)) / P.light_refraction_coefficient;
                const float r_d = sin_theta_d / sqrtf(1.0f - sin_theta_d * sin_theta_d);
                const float factor = r_d / rr;
                ptx *= factor; pty *= factor;
            }
        }
        const float *r = A.rotations + index * 9;                                              // :782-783 (nalgebra gemv: column axpy)
        const float pr0 = ((r[0] * ptx) + r[1] * pty) + r[2];
        const float pr1 = ((r[3] * ptx) + r[4] * pty) + r[5];
        const float pr2 = ((r[6] * ptx) + r[7] * pty) + r[8];
        o = float2{pr0 / pr2, pr1 / pr2};
        if (P.lens_correction_amount < 1.0f) {                                                 // :683-692, :785-851
            GfwLc L;
            L.out_c0 = (float)P.output_width / 2.0f; L.out_c1 = (float)P.output_height / 2.0f;
            L.amount = P.lens_correction_amount;
            L.factor = fmaxf(1.0f - L.amount, 0.001f);
            L.out_f0 = P.f[0] / P.fov / L.factor; L.out_f1 = P.f[1] / P.fov / L.factor;
            L.fov = P.fov;
            o = gfw_lc_solve<MODEL>(L, o.x, o.y, P, C);
        }
    }
    reinterpret_cast<float2 *>(A.out)[i] = o;
}
hipError_t gfw_launch_points(const gfw_kernel_params &P, const GfwCommon &C, const GfwPointsArgs &A, hipStream_t s) {
    const unsigned blocks = (unsigned)((A.n + 255) / 256);
    if (C.model == GFW_MODEL_OPENCV_FISHEYE) hipLaunchKernelGGL(gfw_points_kernel<GFW_MODEL_OPENCV_FISHEYE>, dim3(blocks), dim3(256), 0, s, P, C, A);
    else hipLaunchKernelGGL(gfw_points_kernel<-1>, dim3(blocks), dim3(256), 0, s, P, C, A);
    return hipGetLastError();
}

// Row repack: [rows][14] f32 (FrameTransform.matrices) -> [rows][16] with cos(-m11), sin(-m11) slots.
// Used only for device-resident raw rows (GFW_OPT_MATRICES_ON_DEVICE = 1).  The roll terms are evaluated with
// gfw_cosf / gfw_sinf — the host libm's routines restated (gfw_math.h), bit-identical to what cpu_undistort.rs:159-160
// computes on the host — and, as on the host path (gfw_pack_matrices), only for rows that carry IBIS/OIS data.
__global__ void gfw_repack_matrices_kernel(const float *in, float *out, int rows) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= rows) return;
    const float *m = in + (size_t)r * 14;
    float *o = out + (size_t)r * GFW_MAT_STRIDE;
    float v[14];
    #pragma unroll
    for (int c = 0; c < 14; ++c) { v[c] = m[c]; o[c] = v[c]; }
    float cs = 1.0f, sn = 0.0f;
    if (v[9] != 0.0f || v[10] != 0.0f || v[11] != 0.0f || v[12] != 0.0f || v[13] != 0.0f) { cs = gfw_cosf(-v[11]); sn = gfw_sinf(-v[11]); }
    o[14] = cs; o[15] = sn;
}
hipError_t gfw_launch_repack(const float *in, float *out, int rows, hipStream_t s) {
    hipLaunchKernelGGL(gfw_repack_matrices_kernel, dim3((rows + 255) / 256), dim3(256), 0, s, in, out, rows);
    return hipGetLastError();
}

// 64-bit additive checksum of a device buffer (gfw_checksum64): sum of its little-endian u64 words modulo 2^64, accumulated
// into *out with one atomic per workgroup.  Order-independent, so ranks / launches agree bit for bit.
__global__ __launch_bounds__(256) void gfw_checksum64_kernel(const uint64_t *p, size_t n, unsigned long long *out) {
    __shared__ unsigned long long part[4];
    unsigned long long acc = 0;
    const size_t stride = (size_t)gridDim.x * 256u * 2u;
    const ulonglong2 *p2 = reinterpret_cast<const ulonglong2 *>(p);
    const size_t n2 = n >> 1;
    const size_t step = stride >> 1;
    size_t i = (size_t)blockIdx.x * 256u + threadIdx.x;
    for (; i + 3 * step < n2; i += 4 * step) {        // four independent 16-byte loads in flight per lane
        const ulonglong2 a = p2[i], b = p2[i + step], c = p2[i + 2 * step], d = p2[i + 3 * step];
        acc += (a.x + a.y) + (b.x + b.y) + (c.x + c.y) + (d.x + d.y);
    }
    for (; i < n2; i += step) { const ulonglong2 v = p2[i]; acc += v.x + v.y; }
    if ((n & 1) && blockIdx.x == 0 && threadIdx.x == 0) acc += p[n - 1];
    #pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_down(acc, o, 64);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(out, part[0] + part[1] + part[2] + part[3]);
}
hipError_t gfw_launch_checksum64(const void *buf, size_t bytes, unsigned long long *out, hipStream_t s) {
    const size_t n = bytes / 8;
    if (n == 0) return hipSuccess;
    size_t blocks = (n / 2 + 256 * 8 - 1) / (256 * 8);
    if (blocks < 1) blocks = 1;
    // every workgroup ends with one atomic on the same address, and those serialise across the XCDs: 2048 workgroups spent more time there than
    // reading a 33 MB frame (C5, us per frame with the checksum: 73.4 at 2048 and 1024, 67.8 at 512, 65.5 at 256, 68.0 at 128)
    if (blocks > 256) blocks = 256;
    hipLaunchKernelGGL(gfw_checksum64_kernel, dim3((unsigned)blocks), dim3(256), 0, s, (const uint64_t *)buf, n, out);
    return hipGetLastError();
}

// gfw_set_frame_checksums, behind a launch of the checksum build of the fused kernel: the words of frame f in the launch's table of partial sums
// ([frame][workgroup][wave], every word written by its wave) added to the frame's sum.  GFW_CK_SPLIT workgroups per frame, one word per lane and pass, one atomic per
// workgroup — the first version (ONE workgroup per frame walking 8192 words, 32 dependent passes) took longer than the checksum pass it replaces: C5 50.0 us per
// frame against 53.2, with 42.0 of warp (profiles/r05_c5_checksum.txt).
#define GFW_CK_SPLIT 16
__global__ __launch_bounds__(256) void gfw_ck_finish_kernel(const unsigned long long *part, int per_frame, GfwCkSums S) {
    __shared__ unsigned long long w[4];
    const int f = blockIdx.x / GFW_CK_SPLIT, k = blockIdx.x % GFW_CK_SPLIT;
    const unsigned long long *p = part + (size_t)f * per_frame;
    unsigned long long a0 = 0, a1 = 0;
    int i = k * 256 + threadIdx.x;
    for (; i + GFW_CK_SPLIT * 256 < per_frame; i += 2 * GFW_CK_SPLIT * 256) { a0 += p[i]; a1 += p[i + GFW_CK_SPLIT * 256]; }      // (two loads in flight)
    if (i < per_frame) a0 += p[i];
    unsigned long long acc = a0 + a1;
    #pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_down(acc, o, 64);
    if ((threadIdx.x & 63) == 0) w[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0 && S.sum[f]) atomicAdd(S.sum[f], w[0] + w[1] + w[2] + w[3]);
}
hipError_t gfw_launch_ck_finish(const unsigned long long *part, int per_frame, int n_frames, const GfwCkSums &sums, hipStream_t s) {
    if (n_frames <= 0) return hipSuccess;
    hipLaunchKernelGGL(gfw_ck_finish_kernel, dim3((unsigned)n_frames * GFW_CK_SPLIT), dim3(256), 0, s, part, per_frame, sums);
    return hipGetLastError();
}
// ... and behind every other kernel: the checksum of a written region, byte times 256^(address mod 8).  A lane takes the aligned words of a row in turn: a word
// that lies inside the row whole IS its own contribution; the row's two ends go byte by byte.
__global__ __launch_bounds__(256) void gfw_ck_region_kernel(const uint8_t *dst, long long first_byte, long long stride, int row_bytes, int rows, unsigned long long *out) {
    __shared__ unsigned long long w[4];
    unsigned long long acc = 0;
    for (int r = blockIdx.x; r < rows; r += gridDim.x) {
        const uint8_t *a = dst + first_byte + (long long)r * stride, *e = a + row_bytes;
        const uintptr_t w0 = (uintptr_t)a & ~(uintptr_t)7;
        const int n_words = (int)(((uintptr_t)e + 7 - w0) >> 3);
        for (int j = threadIdx.x; j < n_words; j += 256) {
            const uint8_t *q = (const uint8_t *)(w0 + (uintptr_t)8 * j);
            if (q >= a && q + 8 <= e) acc += *reinterpret_cast<const unsigned long long *>(q);
            else for (int b = 0; b < 8; ++b) if (q + b >= a && q + b < e) acc += (unsigned long long)q[b] << (8 * b);
        }
    }
    #pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_down(acc, o, 64);
    if ((threadIdx.x & 63) == 0) w[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(out, w[0] + w[1] + w[2] + w[3]);
}
hipError_t gfw_launch_ck_region(const uint8_t *dst, long long first_byte, long long stride, int row_bytes, int rows, unsigned long long *out, hipStream_t s) {
    if (rows <= 0 || row_bytes <= 0) return hipSuccess;
    hipLaunchKernelGGL(gfw_ck_region_kernel, dim3((unsigned)(rows < 256 ? rows : 256)), dim3(256), 0, s, dst, first_byte, stride, row_bytes, rows, out);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------- test hooks (gfw_debug_*)
__global__ void gfw_debug_math_kernel(int op, const float *a, const float *b, float *out, size_t n) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float x = a[i], y = b ? b[i] : 1.0f;
    float r = 0.0f;
    switch (op) {
    case 0: r = gfw_atanf(x); break;
    case 1: r = gfw_tanf(x); break;
    case 2: r = gfw_atanf_pos(x); break;
    case 3: r = gfw_div_lean(x, y); break;
    case 4: r = x / y; break;
    case 5: r = gfw_sqrt_lean(x); break;
    case 6: r = sqrtf(x); break;
    case 7: r = (float)gfw_f2i(x); break;
    case 8: r = (float)gfw_f2u_sat(x, 65535.0f); break;
    case 9: r = gfw_round(x); break;
    case 10: r = (float)gfw_f2u_sat(x, 255.0f); break;
    case 11: r = gfw_sinf(x); break;
    case 12: r = gfw_cosf(x); break;
    }
    out[i] = r;
}
hipError_t gfw_launch_debug_math(int op, const float *a, const float *b, float *out, size_t n, hipStream_t s) {
    hipLaunchKernelGGL(gfw_debug_math_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, op, a, b, out, n);
    return hipGetLastError();
}

__device__ __forceinline__ uint64_t gfw_mix64(uint64_t x) {
    x += 0x9E3779B97F4A7C15ull; x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull; x = (x ^ (x >> 27)) * 0x94D049BB133111EBull; return x ^ (x >> 31);
}
__global__ void gfw_debug_selftest_kernel(int test, unsigned long long n, unsigned long long seed, unsigned long long *bad) {
    const unsigned long long stride = (unsigned long long)gridDim.x * blockDim.x;
    unsigned long long local = 0;
    for (unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const uint64_t h = gfw_mix64(i ^ seed);
        if (test == 0) {
            // a: sign, exponent in [2^-19.., 2^19], any mantissa (or exactly 0 once in 64); b: positive, exponent in [2^-20, 2^20]
            const uint32_t ea = 108u + (uint32_t)((h >> 23) % 39u), eb = 107u + (uint32_t)((h >> 55) % 41u);
            float a = gfw_u2f(((uint32_t)(h >> 63) << 31) | (ea << 23) | (uint32_t)(h & 0x7fffffu));
            const float b = gfw_u2f((eb << 23) | (uint32_t)((h >> 29) & 0x7fffffu));
            if (((h >> 47) & 63u) == 0) a = 0.0f;
            const float q = gfw_div_lean(a, b), r = a / b;
            if (gfw_f2u(q) != gfw_f2u(r) && !(q == 0.0f && r == 0.0f)) local++;
        } else if (test == 1) {
            const uint32_t e = 47u + (uint32_t)((h >> 23) % 160u);                    // 2^-80 .. 2^79
            const float x = (((h >> 40) & 255u) == 0) ? 0.0f : gfw_u2f((e << 23) | (uint32_t)(h & 0x7fffffu));
            if (gfw_f2u(gfw_sqrt_lean(x)) != gfw_f2u(sqrtf(x))) local++;
        } else if (test == 2) {
            const float x = gfw_u2f((uint32_t)i & 0x7fffffffu);                       // every non-negative float (incl. inf/NaN)
            const float p = gfw_atanf_pos(x), q = gfw_atanf(x);
            if (gfw_f2u(p) != gfw_f2u(q) && !(p != p && q != q)) local++;
        } else if (test >= 3 && test <= 5) {
            // Exhaustive check of a divide sequence over significand pairs: lane i owns the denominator significand
            // (seed & 0x7fffff) + i and walks numerator significands 0, step, 2*step, ... (step = 1 << (seed >> 32)).
            // Operands in [1, 2) cover both quotient binades; every operation scales exactly with the exponents inside
            // the kernels' checked range, so the significands decide.
            //   3: refined reciprocal + ONE remainder correction (gfw_div_prepared)   4: raw v_rcp + two   5: raw v_rcp + one
            const uint32_t mb = ((uint32_t)seed + (uint32_t)i) & 0x7fffffu;
            const uint32_t step = 1u << (uint32_t)(seed >> 32);
            const float b = gfw_u2f(0x3f800000u | mb);
            const float r0 = gfw_hw_rcp(b);
            const float r = (test == 3) ? __builtin_fmaf(__builtin_fmaf(-b, r0, 1.0f), r0, r0) : r0;
            for (uint32_t ma = 0; ma < (1u << 23); ma += step) {
                const float a = gfw_u2f(0x3f800000u | ma);
                float q = a * r;
                q = __builtin_fmaf(__builtin_fmaf(-b, q, a), r, q);
                if (test == 4) q = __builtin_fmaf(__builtin_fmaf(-b, q, a), r, q);
                if (gfw_f2u(q) != gfw_f2u(a / b)) local++;
            }
        }
    }
    if (local) atomicAdd(bad, local);
}
hipError_t gfw_launch_debug_selftest(int test, unsigned long long n, unsigned long long seed, unsigned long long *bad, hipStream_t s) {
    hipLaunchKernelGGL(gfw_debug_selftest_kernel, dim3(256 * 8), dim3(256), 0, s, test, n, seed, bad);
    return hipGetLastError();
}
