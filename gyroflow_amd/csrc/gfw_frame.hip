// gfw_frame.hip — fused YUV frame kernel: all planes of one frame in ONE launch.
//
// The reference warps a frame plane by plane (src/rendering/mod.rs:655-658), re-deriving the source
// coordinate of every chroma site from scratch although — with its own arithmetic — the chroma site (xc, yc)
// of a plane subsampled by (DW, DH) evaluates `undistort_coord` at exactly the luma position (DW*xc, DH*yc):
//   map_coord(xc, 0, ow/DW, 0, ow) = xc*ow/(ow/DW) = DW*xc       (exact in f32 when xc*ow is exact; checked on host)
// So one thread owns a DW x DH block of luma pixels plus the chroma site that shares the block's top-left
// coordinate: undistort_coord runs once per luma pixel and never for chroma (2x fewer evaluations for 4:2:2,
// 1.5x for 4:2:0, 3x for 4:4:4), the U and V planes share one set of tap weights, and the per-plane
// source_rect map (cpu_undistort.rs:511-514) is applied per plane with a validated divide-by-constant.
//
// Arithmetic is the reference's operation sequence (cpu_undistort.rs:133-228, :421-517, :371-418,
// opencv_fisheye.rs:72-95) with the range scaffolding of divide/sqrt removed (gfw_fastmath.h); operands outside
// the proven range take the generic IEEE path in a (practically never taken) side branch.  Output is
// bit-identical to running gfw_plane_kernel once per plane; tests/test_gpu_parity.py checks both against the
// oracle.
//
// Eligibility (decided on the host, gfw_api.hip): bilinear, background_mode 0, no input rotation,
// lens_correction_amount >= 1, no refraction / mesh / digital lens / IBIS terms / colour-range fix,
// translation3d == 0, stretches in {<=0.001, 1}, full-plane rects, Luma8/Luma16 (+UV8/UV16) planes.
#include <hip/hip_runtime.h>
#include "gfw_warp.h"
#include "gfw_fastmath.h"
#include "gfw_frame.h"

namespace {

struct IeeeOps {
    static __device__ __forceinline__ void div2(float a1, float a2, float b, float &q1, float &q2) { q1 = a1 / b; q2 = a2 / b; }
    static __device__ __forceinline__ float div(float a, float b) { return a / b; }
    static __device__ __forceinline__ float sqrt(float x) { return sqrtf(x); }
    static __device__ __forceinline__ float atan_pos(float x) { return gfw_atanf(x); }
};
struct LeanOps {
    static __device__ __forceinline__ void div2(float a1, float a2, float b, float &q1, float &q2) {
        const GfwRcp d = gfw_rcp_prepare(b);
        q1 = gfw_div_prepared(a1, d); q2 = gfw_div_prepared(a2, d);
    }
    static __device__ __forceinline__ float div(float a, float b) { return gfw_div_lean(a, b); }
    static __device__ __forceinline__ float sqrt(float x) {
        if (__builtin_expect(x < 8.271806125530277e-25f && x != 0.0f, 0)) return sqrtf(x);    // below 2^-80: generic path
        return gfw_sqrt_lean(x);
    }
    static __device__ __forceinline__ float atan_pos(float x) { return gfw_atanf_pos(x); }
};

// opencv_fisheye.rs:72-95 on (X/W, Y/W); then *f, +c (cpu_undistort.rs:155,167)
template <class Ops>
__device__ __forceinline__ void fisheye_project(float X, float Y, float W, const GfwYuvArgs &A, float &u, float &v) {
    float a, b;
    Ops::div2(X, Y, W, a, b);
    if (!A.k_all_zero) {
        const float r = Ops::sqrt(a * a + b * b);
        const float t = Ops::atan_pos(r);
        const float t2 = t * t, t4 = t2 * t2, t6 = t4 * t2, t8 = t4 * t4;
        const float td = t * (1.0f + A.k[0] * t2 + A.k[1] * t4 + A.k[2] * t6 + A.k[3] * t8);
        const float s = (r == 0.0f) ? 1.0f : Ops::div(td, r);
        a = a * s; b = b * s;
    }
    u = a * A.f[0] + A.c[0];
    v = b * A.f[1] + A.c[1];
}

// rotate_and_distort restricted to the eligible configuration (no IBIS/mesh/digital/refraction, t3d == 0).
template <int MODEL>
__device__ __forceinline__ GfwPt rd(float px, float py, int idx, const GfwYuvArgs &A) {
    const float *m = A.matrices + (size_t)idx * GFW_MAT_STRIDE;
    const float4 ma = *reinterpret_cast<const float4 *>(m);
    const float4 mb = *reinterpret_cast<const float4 *>(m + 4);
    const float m8 = m[8];
    const float X = (px * ma.x) + (py * ma.y) + ma.z;
    const float Y = (px * ma.w) + (py * mb.x) + mb.y;
    const float W = (px * mb.z) + (py * mb.w) + m8;
    GfwPt o{0.0f, 0.0f, false};
    if (!(W > 0.0f)) return o;
    if (A.r_limit_sq > 0.0f && (X * X + Y * Y) > A.r_limit_sq * W) return o;
    o.ok = true;
    if (MODEL == GFW_MODEL_OPENCV_FISHEYE) {
        // proven operand range of the lean divide: |X|,|Y| <= 2^19, W in [2^-20, 2^20]  (=> |a|,|b| <= 2^39)
        const float mag = fmaxf(fmaxf(fabsf(X), fabsf(Y)), W);
        const bool lean = (mag <= 524288.0f) && (W >= 9.5367431640625e-07f);
        if (__builtin_expect(lean, 1)) fisheye_project<LeanOps>(X, Y, W, A, o.x, o.y);
        else fisheye_project<IeeeOps>(X, Y, W, A, o.x, o.y);
    } else {
        float du, dv;
        gfw_lens::distort<MODEL>(A.model, X, Y, W, A.kp, A.common, du, dv);
        o.x = du * A.f[0] + A.c[0];
        o.y = dv * A.f[1] + A.c[1];
    }
    if (A.hstretch_div) o.x /= A.hstretch;          // cpu_undistort.rs:222-223 (1.0 and <= 0.001 are skipped on the host)
    if (A.vstretch_div) o.y /= A.vstretch;
    return o;
}

// undistort_coord (cpu_undistort.rs:421-483) for the eligible configuration; (px, py) are full-res output pixels.
template <int MODEL>
__device__ __forceinline__ GfwPt coord(float px, float py, const GfwYuvArgs &A) {
    const float ox = px + A.t2[0], oy = py + A.t2[1];
    const int lim = A.hrs ? A.width : A.height;
    int sy = gfw_f2i(roundf(A.hrs ? ox : oy));
    sy = max(min(sy, lim), 0);
    if (A.matrix_count > 1) {
        const GfwPt pt = rd<MODEL>(ox, oy, A.matrix_count >> 1, A);
        if (pt.ok) { sy = gfw_f2i(roundf(A.hrs ? pt.x : pt.y)); sy = max(min(sy, lim), 0); }
    }
    return rd<MODEL>(ox, oy, min(sy, A.matrix_count - 1), A);
}

// Bilinear taps of an N-channel u8/u16 plane (cpu_undistort.rs:371-418 with I = 2).
template <typename T, int N>
__device__ __forceinline__ void sample_store(float u, float v, const GfwYuvPlane &P, int ox, int oy, bool ok) {
    float out[N];
    #pragma unroll
    for (int c = 0; c < N; ++c) out[c] = P.bg[c];
    if (ok) {
        const int sx0 = gfw_f2i(roundf(u * 32.0f)), sy0 = gfw_f2i(roundf(v * 32.0f));
        const int sx = sx0 >> 5, sy = sy0 >> 5;
        const float cx1 = (float)(sx0 & 31) * 0.03125f, cx0 = 1.0f - cx1;
        const float cy1 = (float)(sy0 & 31) * 0.03125f, cy0 = 1.0f - cy1;
        const T *row0 = reinterpret_cast<const T *>(P.src + (int64_t)sy * P.src_stride) + (int64_t)sx * N;
        const T *row1 = reinterpret_cast<const T *>(reinterpret_cast<const uint8_t *>(row0) + P.src_stride);
        float p00[N], p01[N], p10[N], p11[N];
        if ((unsigned)sx < (unsigned)(P.w - 1) && (unsigned)sy < (unsigned)(P.h - 1)) {      // all four taps inside
            #pragma unroll
            for (int c = 0; c < N; ++c) { p00[c] = (float)row0[c]; p01[c] = (float)row0[N + c]; p10[c] = (float)row1[c]; p11[c] = (float)row1[N + c]; }
        } else {
            const bool x0in = sx >= 0 && sx < P.w, x1in = sx + 1 >= 0 && sx + 1 < P.w;
            const bool y0in = sy >= 0 && sy < P.h, y1in = sy + 1 >= 0 && sy + 1 < P.h;
            #pragma unroll
            for (int c = 0; c < N; ++c) {
                p00[c] = (y0in && x0in) ? (float)row0[c] : P.bg[c];
                p01[c] = (y0in && x1in) ? (float)row0[N + c] : P.bg[c];
                p10[c] = (y1in && x0in) ? (float)row1[c] : P.bg[c];
                p11[c] = (y1in && x1in) ? (float)row1[N + c] : P.bg[c];
            }
            // rows outside the source rect contribute bg*cy (cpu_undistort.rs:408); identical to the tap form only
            // through the same operations, so replay them exactly:
            #pragma unroll
            for (int c = 0; c < N; ++c) {
                float sum = 0.0f;
                if (y0in) { float xs = 0.0f; xs = xs + p00[c] * cx0; xs = xs + p01[c] * cx1; sum = sum + xs * cy0; } else sum = sum + P.bg[c] * cy0;
                if (y1in) { float xs = 0.0f; xs = xs + p10[c] * cx0; xs = xs + p11[c] * cx1; sum = sum + xs * cy1; } else sum = sum + P.bg[c] * cy1;
                out[c] = fminf(sum, P.limit);
            }
            goto store;
        }
        #pragma unroll
        for (int c = 0; c < N; ++c) {
            // xs = 0 + p0*cx0 + p1*cx1 ; sum = 0 + xs0*cy0 + xs1*cy1   (0 + x is exact; taps are non-negative)
            const float xs0 = p00[c] * cx0 + p01[c] * cx1;
            const float xs1 = p10[c] * cx0 + p11[c] * cx1;
            out[c] = fminf(xs0 * cy0 + xs1 * cy1, P.limit);
        }
    }
store:
    T *dst = reinterpret_cast<T *>(P.dst + (int64_t)oy * P.dst_stride) + (int64_t)ox * N;
    #pragma unroll
    for (int c = 0; c < N; ++c) dst[c] = (T)gfw_f2u_sat(out[c], sizeof(T) == 1 ? 255.0f : 65535.0f);
}

// ---- first pass (rolling-shutter row pick) -----------------------------------------------------------------
// The mid-row projection of undistort_coord (cpu_undistort.rs:470-479) is used for ONE thing: the integer
// sy = clamp(round(p.y)).  FAST1 evaluates p.y with fused arithmetic and a per-lens table of
// s(rho) = theta_d(atan(sqrt(rho)))/sqrt(rho) (linear interpolation, rho = (X/W)^2 + (Y/W)^2) and accepts the
// rounded value only when no half-integer lies within +-E of it, E bounding |approx - exact| (derivation in
// DESIGN.md section 2; tests/test_gpu_pass1.py measures the real gap).  Everything else — a few percent of the
// pixels — goes through the exact projection: queued in LDS and processed densely by the whole workgroup, so
// the exact path costs one wave-pass per tile instead of one per wave.
#define GFW_QCAP 256
struct Pass1 { float X, Y, W; };

template <int MODEL, bool FAST1>
__device__ __forceinline__ int pass1_default_row(float ox, float oy, const GfwYuvArgs &A) {
    const int lim = A.hrs ? A.width : A.height;
    const int sy = gfw_f2i(roundf(A.hrs ? ox : oy));
    return max(min(sy, lim), 0);
}
// exact: cpu_undistort.rs:470-479
template <int MODEL>
__device__ __forceinline__ int pass1_exact(float ox, float oy, const GfwYuvArgs &A) {
    int sy = pass1_default_row<MODEL, false>(ox, oy, A);
    const GfwPt pt = rd<MODEL>(ox, oy, A.matrix_count >> 1, A);
    if (pt.ok) { const int lim = A.hrs ? A.width : A.height; sy = gfw_f2i(roundf(A.hrs ? pt.x : pt.y)); sy = max(min(sy, lim), 0); }
    return sy;
}
// approximate + certificate; returns false when the exact path must decide
__device__ __forceinline__ bool pass1_fast(float ox, float oy, const GfwYuvArgs &A, const float *mid, const float2 *tab, int &sy) {
    // X, Y, W with fused multiply-adds (|error| ~ 1e-7 relative to the term magnitudes)
    const float X = __builtin_fmaf(ox, mid[0], __builtin_fmaf(oy, mid[1], mid[2]));
    const float Y = __builtin_fmaf(ox, mid[3], __builtin_fmaf(oy, mid[4], mid[5]));
    const float W = __builtin_fmaf(ox, mid[6], __builtin_fmaf(oy, mid[7], mid[8]));
    if (!(W > 0.0009765625f)) return false;                        // W not safely positive: exact path decides validity
    const float rw = gfw_hw_rcp(W);
    const float a = X * rw, b = Y * rw;
    const float rho = __builtin_fmaf(a, a, b * b);
    if (!(rho < A.p1_rho_max)) return false;                       // outside the table (or NaN)
    if (A.r_limit_sq > 0.0f) {                                     // :139 — decide only when clear of the boundary
        const float lhs = __builtin_fmaf(X, X, Y * Y), rhs = A.r_limit_sq * W;
        if (!(lhs < rhs * 0.9999f)) return false;
    }
    const float tpos = rho * A.p1_rho_scale;
    const float ti = floorf(tpos);
    const float2 e = tab[(int)ti];
    const float s = __builtin_fmaf(tpos - ti, e.y, e.x);
    const float v = __builtin_fmaf((A.hrs ? a : b) * s, A.p1_f, A.p1_c);
    const float g = v - 0.5f;
    const float dist = fabsf(g - rintf(g));                        // distance of v to the nearest half-integer
    const float lim = (float)(A.hrs ? A.width : A.height);
    const bool inside = v > -0.25f && v < lim + 0.25f;             // outside, the clamp decides and ties cannot matter
    if (inside && !(dist > A.p1_eps)) return false;
    sy = max(min(gfw_f2i(rintf(v)), (int)lim), 0);
    return true;
}

template <int MODEL, typename T, int DW, int DH, bool INTERLEAVED_UV, int RB, bool FAST1>
__global__ __launch_bounds__(256) void gfw_yuv_kernel(const GfwYuvArgs A) {
    // tile = 64 x 4 lanes, each lane owns RB vertically stacked DW x DH luma blocks (+ their chroma sites)
    __shared__ float2 s_tab[FAST1 ? GFW_P1_TABLE_N + 1 : 1];
    __shared__ float q_x[FAST1 ? GFW_QCAP : 1], q_y[FAST1 ? GFW_QCAP : 1];
    __shared__ int q_sy[FAST1 ? GFW_QCAP : 1];
    __shared__ unsigned q_n;
    const int tid = threadIdx.y * 64 + threadIdx.x;
    const bool two_pass = A.matrix_count > 1;
    if (FAST1 && two_pass) {
        for (int i = tid; i <= GFW_P1_TABLE_N; i += 256) s_tab[i] = A.p1_table[i];
        if (tid == 0) q_n = 0;
        __syncthreads();
    }
    const int tiles_x = A.tiles_x;
    const int b = blockIdx.x;
    const int n = tiles_x * A.tiles_y;
    const int per = (n + 7) >> 3;
    const int t = (b & 7) * per + (b >> 3);          // XCD-banded tile order (workgroup b runs on XCD b % 8)
    const bool tile_ok = t < n;
    const int ty = tile_ok ? t / tiles_x : 0, tx = tile_ok ? t - ty * tiles_x : 0;
    const int cx = tx * 64 + threadIdx.x;
    const int cy0 = (ty * 4 + threadIdx.y) * RB;     // first chroma-site row of this lane
    const bool lane_ok = tile_ok && cx < A.cw;

    // ---- phase 1: rolling-shutter row of every luma pixel of this lane --------------------------------
    int rows[RB][DH][DW];
    if (two_pass) {
        const float *mid = A.matrices + (size_t)(A.matrix_count >> 1) * GFW_MAT_STRIDE;   // wave-uniform -> scalar loads
        #pragma unroll
        for (int r = 0; r < RB; ++r) {
            #pragma unroll
            for (int j = 0; j < DH; ++j) {
                #pragma unroll
                for (int i = 0; i < DW; ++i) {
                    const int lx = cx * DW + i, ly = (cy0 + r) * DH + j;
                    int sy = 0;
                    if (lane_ok && lx < A.out_w && ly < A.out_h) {
                        const float ox = (float)lx + A.t2[0], oy = (float)ly + A.t2[1];
                        if (FAST1) {
                            if (!pass1_fast(ox, oy, A, mid, s_tab, sy)) {
                                const unsigned slot = atomicAdd(&q_n, 1u);
                                if (slot < GFW_QCAP) { q_x[slot] = ox; q_y[slot] = oy; sy = -1 - (int)slot; }
                                else sy = pass1_exact<MODEL>(ox, oy, A);          // queue full: decide inline
                                if (A.audit) atomicAdd(&A.audit[slot < GFW_QCAP ? 2 : 3], 1ull);
                            } else if (A.audit) {                                 // audit mode: every certificate is checked
                                atomicAdd(&A.audit[0], 1ull);
                                if (pass1_exact<MODEL>(ox, oy, A) != sy) atomicAdd(&A.audit[1], 1ull);
                            }
                        } else {
                            sy = pass1_exact<MODEL>(ox, oy, A);
                        }
                    }
                    rows[r][j][i] = sy;
                }
            }
        }
        if (FAST1) {
            // ---- phase 2: the workgroup resolves the queued pixels exactly, densely packed ---------------
            __syncthreads();
            const unsigned qn = min(q_n, (unsigned)GFW_QCAP);
            for (unsigned e = tid; e < qn; e += 256) q_sy[e] = pass1_exact<MODEL>(q_x[e], q_y[e], A);
            __syncthreads();
        }
    }
    if (!lane_ok) return;

    // ---- phase 3: exact projection with the row's own matrix, then taps -------------------------------
    #pragma unroll
    for (int r = 0; r < RB; ++r) {
        const int cy = cy0 + r;
        if (cy >= A.ch) break;
        float u0 = 0.0f, v0 = 0.0f; bool ok0 = false;
        #pragma unroll
        for (int j = 0; j < DH; ++j) {
            #pragma unroll
            for (int i = 0; i < DW; ++i) {
                const int lx = cx * DW + i, ly = cy * DH + j;
                if (lx >= A.out_w || ly >= A.out_h) continue;
                const float ox = (float)lx + A.t2[0], oy = (float)ly + A.t2[1];
                int sy;
                if (two_pass) { sy = rows[r][j][i]; if (FAST1 && sy < 0) sy = q_sy[-1 - sy]; }
                else sy = pass1_default_row<MODEL, false>(ox, oy, A);
                const GfwPt p = rd<MODEL>(ox, oy, min(sy, A.matrix_count - 1), A);
                if (i == 0 && j == 0) { u0 = p.x; v0 = p.y; ok0 = p.ok; }
                const float lu = gfw_map_const(p.x, A.map_lx), lv = gfw_map_const(p.y, A.map_ly);   // cpu_undistort.rs:511-514
                sample_store<T, 1>(lu, lv, A.pl[0], lx, ly, p.ok);
            }
        }
        if (A.nplanes > 1) {
            const float cu = gfw_map_const(u0, A.map_cx), cv = gfw_map_const(v0, A.map_cy);
            if (INTERLEAVED_UV) {
                sample_store<T, 2>(cu, cv, A.pl[1], cx, cy, ok0);
            } else {
                sample_store<T, 1>(cu, cv, A.pl[1], cx, cy, ok0);
                if (A.nplanes > 2) sample_store<T, 1>(cu, cv, A.pl[2], cx, cy, ok0);
                if (A.nplanes > 3) sample_store<T, 1>(cu, cv, A.pl[3], cx, cy, ok0);
            }
        }
    }
}

template <int MODEL, typename T, int RB, bool FAST1>
hipError_t launch_mt(const GfwYuvArgs &A, int dw, int dh, bool interleaved, hipStream_t s) {
    const int grid = (((A.tiles_x * A.tiles_y) + 7) >> 3) << 3;
    if (grid <= 0) return hipSuccess;
    dim3 block(64, 4);
#define GFW_YUV_LAUNCH(DW, DH, IL) hipLaunchKernelGGL((gfw_yuv_kernel<MODEL, T, DW, DH, IL, RB, FAST1>), dim3(grid), block, 0, s, A)
    if (dw == 2 && dh == 1 && !interleaved) GFW_YUV_LAUNCH(2, 1, false);
    else if (dw == 2 && dh == 1 && interleaved) GFW_YUV_LAUNCH(2, 1, true);
    else if (dw == 2 && dh == 2 && !interleaved) GFW_YUV_LAUNCH(2, 2, false);
    else if (dw == 2 && dh == 2 && interleaved) GFW_YUV_LAUNCH(2, 2, true);
    else if (dw == 1 && dh == 1 && !interleaved) GFW_YUV_LAUNCH(1, 1, false);
    else if (dw == 1 && dh == 1 && interleaved) GFW_YUV_LAUNCH(1, 1, true);
    else return hipErrorInvalidValue;
#undef GFW_YUV_LAUNCH
    return hipGetLastError();
}

}  // namespace

int gfw_yuv_rows_per_lane(bool fast1) { return fast1 ? GFW_YUV_RB_FAST : GFW_YUV_RB_EXACT; }

hipError_t gfw_launch_yuv(const GfwYuvArgs &A, int bytes_per_sample, int dw, int dh, bool interleaved, bool fast1, hipStream_t s) {
    if (A.model == GFW_MODEL_OPENCV_FISHEYE) {
        if (fast1)
            return bytes_per_sample == 1 ? launch_mt<GFW_MODEL_OPENCV_FISHEYE, uint8_t, GFW_YUV_RB_FAST, true>(A, dw, dh, interleaved, s)
                                         : launch_mt<GFW_MODEL_OPENCV_FISHEYE, uint16_t, GFW_YUV_RB_FAST, true>(A, dw, dh, interleaved, s);
        return bytes_per_sample == 1 ? launch_mt<GFW_MODEL_OPENCV_FISHEYE, uint8_t, GFW_YUV_RB_EXACT, false>(A, dw, dh, interleaved, s)
                                     : launch_mt<GFW_MODEL_OPENCV_FISHEYE, uint16_t, GFW_YUV_RB_EXACT, false>(A, dw, dh, interleaved, s);
    }
    return bytes_per_sample == 1 ? launch_mt<-1, uint8_t, GFW_YUV_RB_EXACT, false>(A, dw, dh, interleaved, s)
                                 : launch_mt<-1, uint16_t, GFW_YUV_RB_EXACT, false>(A, dw, dh, interleaved, s);
}
