// gfw_frame.hip — fused YUV frame kernel: all planes of one frame in ONE launch.
//
// The reference warps a frame plane by plane (src/rendering/mod.rs:655-658), re-deriving the source
// coordinate of every chroma site from scratch although — with its own arithmetic — the chroma site (xc, yc)
// of a plane subsampled by (DW, DH) evaluates `undistort_coord` at exactly the luma position (DW*xc, DH*yc):
//   map_coord(xc, 0, ow/DW, 0, ow) = xc*ow/(ow/DW) = DW*xc       (exact in f32 when xc*ow is exact; checked on host)
// So one lane owns DW x DH luma pixels plus the chroma site that shares the block's top-left coordinate:
// undistort_coord runs once per luma pixel and never for chroma (2x fewer evaluations for 4:2:2, 1.5x for
// 4:2:0, 3x for 4:4:4), U and V share bins, weights and offsets, and the per-plane source_rect map
// (cpu_undistort.rs:511-514) is a validated divide-by-constant.
//
// Arithmetic is the reference's operation sequence (cpu_undistort.rs:133-228, :421-517, :371-418,
// opencv_fisheye.rs:72-95) with the range scaffolding of divide/sqrt removed (gfw_fastmath.h); operands outside
// the proven range take the generic IEEE path in a (practically never taken) side branch.  Output is
// bit-identical to running gfw_plane_kernel once per plane; tests/test_gpu_parity.py checks both against the
// oracle.
//
// Shape of the kernel (all of it driven by measurements in profiles/):
//   * VALU-issue bound, so instruction count and code size are what matter: every code path exists once (row
//     loop not unrolled) and slow paths are side branches — a fully unrolled 26 K-instruction body measured slower
//     (instruction cache).  Pinning the uniform floats in VGPRs (GFW_PIN_UNIFORMS) was tried and rejected: it cost
//     occupancy (100 VGPRs) and ran 15 % slower than leaving them to the scalar file;
//   * persistent workgroups: the grid is sized to the machine and each workgroup walks a band of tiles, so the
//     ~5 us start-up of a wave (kernel-argument loads) is paid once, not once per tile;
//   * XCD-banded tile order (workgroup b runs on XCD b % 8): an XCD's L2 sees a contiguous band of source lines.
//
// Eligibility (decided on the host, gfw_api.hip build_yuv_args): bilinear / bicubic / Lanczos4 taps (this file is
// compiled once per tap count and sample type), background_mode 0-2, no input rotation, lens_correction_amount >= 1,
// no mesh / colour-range fix / fill flag, translation3d == 0 (refraction, digital lens and IBIS/OIS terms are served by
// the generic-model instantiation with the exact first pass), stretches in
// {<=0.001, 1}, full-plane rects; Luma8/Luma16 (+UV8/UV16) planes with chroma planes of identical geometry, one
// packed RGB(A)8/16 / BGRA8 / AYUV16 / RGBAf plane, or planar R32f planes.
#include <hip/hip_runtime.h>
#include "gfw_warp.h"
#include "gfw_fastmath.h"
#include "gfw_frame.h"

// measured switches (1 = on): branch-free rounding, exact-FMA row sums, hardware min for the limit clamp
#ifndef GFW_EXP_ROUND
#define GFW_EXP_ROUND 1
#endif
#ifndef GFW_EXP_FMA
#define GFW_EXP_FMA 1
#endif
#ifndef GFW_EXP_MIN
#define GFW_EXP_MIN 1
#endif
#ifndef GFW_TAP_ROW_UNROLL8
#define GFW_TAP_ROW_UNROLL8 1     // tap rows in flight in the Lanczos4 path (measured: 1 beats 2)
#endif
#ifndef GFW_TAP_ROW_UNROLL
#define GFW_TAP_ROW_UNROLL 2      // tap rows fetched together by the bicubic / Lanczos4 paths (registers vs loads in flight)
#endif

namespace {

// 32-phase bicubic / Lanczos4 tap table (one constant copy per translation unit)
__device__
#include "gfw_coeffs.inc"


// A wave-uniform float pinned in a VGPR (keeps SGPRs for pointers / exec masks; VALU reads either at no cost).
#ifndef GFW_PIN_UNIFORMS
#define GFW_PIN_UNIFORMS 0
#endif
__device__ __forceinline__ float vu(float s) {
#if GFW_PIN_UNIFORMS
    float v; asm volatile("v_mov_b32 %0, %1" : "=v"(v) : "s"(s)); return v;
#else
    return s;
#endif
}

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
    // glibc atanf, wave-specialised: when every active lane is below 0.4375 the reduction (and its division)
    // disappears; otherwise the select-based single-division form (gfw_fastmath.h).
    static __device__ __forceinline__ float atan_pos(float x) {
        if (__all(x < 0.4375f)) {
            const float aT0 = 3.3333334327e-01f, aT1 = -2.0000000298e-01f, aT2 = 1.4285714924e-01f, aT3 = -1.1111110449e-01f,
                        aT4 = 9.0908870101e-02f, aT5 = -7.6918758452e-02f, aT6 = 6.6610731184e-02f, aT7 = -5.8335702866e-02f,
                        aT8 = 4.9768779427e-02f, aT9 = -3.6531571299e-02f, aT10 = 1.6285819933e-02f;
            const float z = x * x, w = z * z;
            const float s1 = z * (aT0 + w * (aT2 + w * (aT4 + w * (aT6 + w * (aT8 + w * aT10)))));
            const float s2 = w * (aT1 + w * (aT3 + w * (aT5 + w * (aT7 + w * aT9))));
            return x - x * (s1 + s2);
        }
        return gfw_atanf_pos(x);
    }
};

// Lens + per-plane uniforms, VGPR-resident for the pixel loops.
struct Lens {
    float f0, f1, c0, c1, k0, k1, k2, k3, t2x, t2y, rl2;
};
struct Maps {                       // source_rect maps: u * mul / den  (den, rcp shared by luma and chroma)
    float mul_lx, mul_ly, mul_cx, mul_cy, den_x, rcp_x, den_y, rcp_y;
};
// INF_SAFE: an infinite coordinate must stay infinite (x*mul/den in IEEE; the reference then casts it to i32::MIN/MAX and
// reads background), but the remainder step turns it into inf - inf = NaN; clamping the NaN remainder to a finite value
// restores q0's infinity and changes nothing for finite or NaN inputs.  The specialised fisheye projection cannot produce
// an infinite coordinate (a*s is bounded by theta_d), so only the generic-model instantiation pays for the guard.
template <bool INF_SAFE>
__device__ __forceinline__ float map_c(float x, float mul, float den, float rcp) {
    const float a = x * mul;
    const float q0 = a * rcp;
    float r0 = __builtin_fmaf(-den, q0, a);
    if (INF_SAFE) r0 = fmaxf(r0, -3.4028234664e38f);
    return __builtin_fmaf(r0, rcp, q0);
}

// opencv_fisheye.rs:72-95 on (X/W, Y/W); then *f, +c (cpu_undistort.rs:155,167)
template <class Ops>
__device__ __forceinline__ void fisheye_project(float X, float Y, float W, const Lens &L, bool k_all_zero, float &u, float &v) {
    float a, b;
    Ops::div2(X, Y, W, a, b);
    if (!k_all_zero) {
        const float r = Ops::sqrt(a * a + b * b);
        const float t = Ops::atan_pos(r);
        const float t2 = t * t, t4 = t2 * t2, t6 = t4 * t2, t8 = t4 * t4;
        const float td = t * (1.0f + L.k0 * t2 + L.k1 * t4 + L.k2 * t6 + L.k3 * t8);
        const float s = (r == 0.0f) ? 1.0f : Ops::div(td, r);
        a = a * s; b = b * s;
    }
    u = a * L.f0 + L.c0;
    v = b * L.f1 + L.c1;
}

// rotate_and_distort (cpu_undistort.rs:133-228) restricted to the eligible configuration
// (no mesh, translation3d == 0; IBIS terms, digital lens and refraction only through the generic-model instantiation).  ma/mb/m8 = the 9 matrix entries of the chosen row.
template <int MODEL>
__device__ __forceinline__ GfwPt rd(float px, float py, const float4 ma, const float4 mb, const float m8, const float *ext, const Lens &L, const GfwYuvArgs &A) {
    const float X = (px * ma.x) + (py * ma.y) + ma.z;
    const float Y = (px * ma.w) + (py * mb.x) + mb.y;
    const float W = (px * mb.z) + (py * mb.w) + m8;
    GfwPt o{0.0f, 0.0f, false};
    if (!(W > 0.0f)) return o;
    if (L.rl2 > 0.0f && (X * X + Y * Y) > L.rl2 * W) return o;
    o.ok = true;
    if (MODEL == GFW_MODEL_OPENCV_FISHEYE) {
        // proven operand range of the lean divide: |X|,|Y| <= 2^19, W in [2^-20, 2^20]  (=> |a|,|b| <= 2^39)
        const float mag = fmaxf(fmaxf(fabsf(X), fabsf(Y)), W);
        const bool lean = (mag <= 524288.0f) && (W >= 9.5367431640625e-07f);
        if (__builtin_expect(lean, 1)) fisheye_project<LeanOps>(X, Y, W, L, A.k_all_zero != 0, o.x, o.y);
        else fisheye_project<IeeeOps>(X, Y, W, L, A.k_all_zero != 0, o.x, o.y);      // generic IEEE expansions
    } else {
        // every lens model through the generic IEEE routines, plus the optional stages of rotate_and_distort in the
        // reference's order: refraction (:143-152), model, *f, IBIS/OIS rotate + shift (:157-165), +c, digital lens (:216-220)
        float Wd = W;
        if ((A.extras & 4) && W != 0.0f) {
            const float r = sqrtf(X * X + Y * Y) / W;
            const float sin_theta_d = (r / sqrtf(1.0f + r * r)) * A.kp.light_refraction_coefficient;
            const float r_d = sin_theta_d / sqrtf(1.0f - sin_theta_d * sin_theta_d);
            if (r_d != 0.0f) Wd *= r / r_d;
        }
        float du, dv;
        gfw_lens::distort<MODEL>(A.model, X, Y, Wd, A.kp, A.common, du, dv);
        float u = du * L.f0, v = dv * L.f1;
        if (A.extras & 1) {
            const float m9 = ext[1], m10 = ext[2], m11 = ext[3], m12 = ext[4], m13 = ext[5];
            if (m9 != 0.0f || m10 != 0.0f || m11 != 0.0f || m12 != 0.0f || m13 != 0.0f) {
                const float cos_a = ext[6], sin_a = ext[7];               // cosf(-m11), sinf(-m11) from the host libm
                const float nu = cos_a * u - sin_a * v - m9 + m12;
                const float nv = sin_a * u + cos_a * v - m10 + m13;
                u = nu; v = nv;
            }
        }
        u = u + L.c0; v = v + L.c1;
        if (A.extras & 2) {
            float d0, d1;
            gfw_lens::distort<-1>(A.common.digital, u, v, 1.0f, A.kp, A.common, d0, d1);
            u = d0; v = d1;
        }
        o.x = u; o.y = v;
    }
    // input_{horizontal,vertical}_stretch (cpu_undistort.rs:222-223): only <= 0.001 (skipped) or 1.0 (x/1 == x) reach
    // this kernel; any other value is routed to the per-plane kernel by the host.
    return o;
}
template <int MODEL>
__device__ __forceinline__ GfwPt rd_row(float px, float py, int idx, const Lens &L, const GfwYuvArgs &A) {
    const float *m = A.matrices + (size_t)idx * GFW_MAT_STRIDE;
    return rd<MODEL>(px, py, *reinterpret_cast<const float4 *>(m), *reinterpret_cast<const float4 *>(m + 4), m[8], m + 8, L, A);
}

// f32::round (half away from zero) then `as i32`: rndne is exact except on ties, which take the side branch.
// `x.round() as i32` (half away from zero, then truncating saturating cast) without the tie branch:
// trunc(x + copysign(pred(0.5), x)) — equal to the cast of roundf(x) for every one of the 2^32 floats
// (tests/test_math_host.py checks this exhaustively); the cast itself truncates.
__device__ __forceinline__ int round_i32(float x) {
#if GFW_EXP_ROUND
    return gfw_f2i(x + copysignf(0x1.fffffep-2f, x));
#else
    float r = rintf(x);
    if (__builtin_expect(fabsf(x - r) == 0.5f, 0)) r = truncf(x) + copysignf(1.0f, x);
    return gfw_f2i(r);
#endif
}
// f32::min(v, limit) with the hardware's IEEE-mode v_min_f32 (non-NaN operand wins, as Rust's does): spares the
// canonicalising v_max the compiler puts in front of fminf for a uniform operand.
__device__ __forceinline__ float min_limit(float v, float limit) {
    float r; asm("v_min_f32 %0, %1, %2" : "=v"(r) : "v"(v), "s"(limit)); return r;
}

// ---- LUT taps (cpu_undistort.rs:371-418): I = 2 bilinear, 4 bicubic, 8 Lanczos4 --------------------------------
// The I x-weights and I y-weights of a sample stay in the LDS copy of the table (32 phases x I floats per filter); a
// Bins holds the two row pointers.  (Keeping 2*I weights per sample in registers cost the bicubic / Lanczos4 kernels
// their occupancy: 148 VGPRs for I = 8.)
template <int I> struct Bins { int sx, sy; const float *tx, *ty; };
template <int I>
__device__ __forceinline__ Bins<I> make_bins(float u, float v, const float *lut) {
    constexpr float OFFSET = (I == 2) ? 0.0f : (I == 4 ? 1.0f : 3.0f);       // :374
    const int sx0 = round_i32((u - OFFSET) * 32.0f), sy0 = round_i32((v - OFFSET) * 32.0f);
    Bins<I> b;
    b.sx = sx0 >> 5; b.sy = sy0 >> 5;
    constexpr int IND = (I == 2) ? 0 : (I == 4) ? 64 : 192, SHIFT = (I >> 2) + 1;       // :373-375
    b.tx = lut + IND + ((sx0 & 31) << SHIFT); b.ty = lut + IND + ((sy0 & 31) << SHIFT);
    return b;
}
template <typename T> struct is_f32 { static constexpr bool value = false; };
template <> struct is_f32<float> { static constexpr bool value = true; };
// Taps that straddle the source rect: out-of-rect taps read `bg`, out-of-rect rows contribute bg*cy
// (cpu_undistort.rs:391-411), in the reference's exact operation order.
template <typename T, int N, int I>
__device__ __forceinline__ void taps_edge(const uint8_t *src, int stride, const Bins<I> &b, int w, int h, const float *bg, float limit, float *out) {
    float sum[N];
    #pragma unroll
    for (int c = 0; c < N; ++c) sum[c] = 0.0f;
    #pragma unroll 1        // the rare path: rolled, weights read from the table as they are needed
    for (int yp = 0; yp < I; ++yp) {
        const int yy = b.sy + yp;
        const float wy = b.ty[yp];
        if (yy >= 0 && yy < h) {
            const T *row = reinterpret_cast<const T *>(src + (int64_t)yy * stride);
            float xs[N];
            #pragma unroll
            for (int c = 0; c < N; ++c) xs[c] = 0.0f;
            #pragma unroll 1
            for (int xp = 0; xp < I; ++xp) {
                const int xx = b.sx + xp;
                const bool in = xx >= 0 && xx < w;
                const float wx = b.tx[xp];
                #pragma unroll
                for (int c = 0; c < N; ++c) { const float px = in ? (float)row[(int64_t)xx * N + c] : bg[c]; xs[c] = xs[c] + px * wx; }
            }
            #pragma unroll
            for (int c = 0; c < N; ++c) sum[c] = sum[c] + xs[c] * wy;
        } else {
            #pragma unroll
            for (int c = 0; c < N; ++c) sum[c] = sum[c] + bg[c] * wy;
        }
    }
    #pragma unroll
    for (int c = 0; c < N; ++c) out[c] = fminf(sum[c], limit);
}
// All I x I taps inside.  Bilinear on integer pixels: every tap is >= +0, so the reference's leading zero-adds
// (xsum = 0 + p*c, sum = 0 + xs*cy) are exact identities and are dropped; everywhere else (negative weights, f32
// pixels with -0 / negative values) they are kept so that signed zeros come out as the reference's.
template <typename T, int N, int I>
__device__ __forceinline__ void taps_inside(const uint8_t *src, int off0, int stride, const Bins<I> &b, float limit, float *out) {
    float cx[I];
    #pragma unroll
    for (int i = 0; i < I; ++i) cx[i] = b.tx[i];
    if (N == 1 && !is_f32<T>::value && I > 2) {
        // single-channel integer planes (Y, U, V): the I taps of a row are I*sizeof(T) contiguous bytes.  They are fetched as
        // ALIGNED dwords and funnel-shifted into place: a row fetch whose address is not 4-byte aligned (every odd u16
        // pixel) costs the texture-address unit 65 cycles instead of 18 (profiles/r01_membench_tap_row_fetch.txt), and that
        // was what the Lanczos4 kernel waited for.  The extra dword is only read when the row is misaligned; the host-side
        // `inside` test keeps TAP_MARGIN pixels clear of the row end so that it never leaves the plane.
        constexpr int ND = (I * (int)sizeof(T)) / 4;
        float s1 = 0.0f;
        auto row = [&](const uint32_t *wp, unsigned mis, unsigned sh, float wy) {
            uint32_t w[ND + 1];
            #pragma unroll
            for (int j = 0; j < ND; ++j) w[j] = wp[j];
            w[ND] = mis ? wp[ND] : 0u;
            float xs = 0.0f;
            #pragma unroll
            for (int j = 0; j < ND; ++j) {
                const uint32_t d = __builtin_amdgcn_alignbit(w[j + 1], w[j], sh);       // ({w[j+1], w[j]} >> sh)[31:0]
                if (sizeof(T) == 2) {
                    xs = xs + (float)(d & 0xffffu) * cx[2 * j];
                    xs = xs + (float)(d >> 16) * cx[2 * j + 1];
                } else {
                    xs = xs + (float)(d & 0xffu) * cx[4 * j];
                    xs = xs + (float)((d >> 8) & 0xffu) * cx[4 * j + 1];
                    xs = xs + (float)((d >> 16) & 0xffu) * cx[4 * j + 2];
                    xs = xs + (float)(d >> 24) * cx[4 * j + 3];
                }
            }
            s1 = s1 + xs * wy;
        };
        if ((stride & 3) == 0) {
            // the usual case (row pitch a multiple of 4 bytes): the misalignment is the same for every tap row of the sample
            const uint8_t *rp0 = src + (int64_t)off0;
            const unsigned mis = (unsigned)(uintptr_t)rp0 & 3u, sh = mis * 8u;
            const uint8_t *ap = rp0 - mis;
            #pragma unroll (I >= 8 ? GFW_TAP_ROW_UNROLL8 : GFW_TAP_ROW_UNROLL)
            for (int yp = 0; yp < I; ++yp) row(reinterpret_cast<const uint32_t *>(ap + (int64_t)yp * stride), mis, sh, b.ty[yp]);
        } else {
            #pragma unroll 1
            for (int yp = 0; yp < I; ++yp) {
                const uint8_t *rp = src + (int64_t)(off0 + yp * stride);
                const unsigned mis = (unsigned)(uintptr_t)rp & 3u;
                row(reinterpret_cast<const uint32_t *>(rp - mis), mis, mis * 8u, b.ty[yp]);
            }
        }
        out[0] = fminf(s1, limit);
        return;
    }
    float sum[N];
    #pragma unroll
    for (int c = 0; c < N; ++c) sum[c] = 0.0f;
    #pragma unroll
    for (int yp = 0; yp < I; ++yp) {
        const T *row = reinterpret_cast<const T *>(src + (int64_t)(off0 + yp * stride));
        float xs[N];
        #pragma unroll
        for (int c = 0; c < N; ++c) xs[c] = 0.0f;
        #pragma unroll
        for (int xp = 0; xp < I; ++xp) {
            #pragma unroll
            for (int c = 0; c < N; ++c) xs[c] = xs[c] + (float)row[xp * N + c] * cx[xp];
        }
        #pragma unroll
        for (int c = 0; c < N; ++c) sum[c] = sum[c] + xs[c] * b.ty[yp];
    }
    #pragma unroll
    for (int c = 0; c < N; ++c) out[c] = fminf(sum[c], limit);
}
// TAP_MARGIN: pixels kept clear of the row end by the aligned dword fetch of taps_inside (one dword may extend
// 4/sizeof(T) - 1 pixels past the last tap); samples closer to the edge take the exact edge path.
template <typename T, int N, int I>
__device__ __forceinline__ bool bins_inside(const Bins<I> &b, int w, int h) {
    constexpr int TAP_MARGIN = (N == 1 && !is_f32<T>::value && I > 2) ? (4 / (int)sizeof(T) - 1) : 0;
    return w >= I + TAP_MARGIN && h >= I && (unsigned)b.sx <= (unsigned)(w - I - TAP_MARGIN) && (unsigned)b.sy <= (unsigned)(h - I);
}
template <typename T, int N>
__device__ __forceinline__ void store_px(uint8_t *dst, int off, const float *v) {
    T *d = reinterpret_cast<T *>(dst + (int64_t)off);
    #pragma unroll
    for (int c = 0; c < N; ++c) {
        if (is_f32<T>::value) d[c] = (T)v[c];                                   // f32 pixels pass through (pixel_formats.rs:247,296)
        else d[c] = (T)gfw_f2u_sat(v[c], sizeof(T) == 1 ? 255.0f : 65535.0f);    // `as u8/u16`
    }
}
// One plane.  32-bit byte offsets from the uniform plane base (planes are < 2 GiB, checked on the host).
template <typename T, int N, int I>
__device__ __forceinline__ void sample_store(float u, float v, bool ok, const GfwYuvPlane &P, const float *bg, float limit, int ox, int oy, const float *lut) {
    float out[N];
    #pragma unroll
    for (int c = 0; c < N; ++c) out[c] = bg[c];
    if (ok) {
        const Bins<I> b = make_bins<I>(u, v, lut);
        if (__builtin_expect((bins_inside<T, N, I>(b, P.w, P.h)), 1))
            taps_inside<T, N, I>(P.src, b.sy * P.src_stride + b.sx * (int)(N * sizeof(T)), P.src_stride, b, limit, out);
        else
            taps_edge<T, N, I>(P.src, P.src_stride, b, P.w, P.h, bg, limit, out);
    }
    store_px<T, N>(P.dst, oy * P.dst_stride + ox * (int)(N * sizeof(T)), out);
}
// Planar planes of identical geometry (U and V; or G,B,R,A of a planar float frame): one set of bins / weights /
// offsets, one gather set per plane.
template <typename T, int I>
__device__ __forceinline__ void sample_store_shared(float u, float v, bool ok, const GfwYuvPlane *pl, int first, int last, int ox, int oy, const float *lut) {
    const GfwYuvPlane &P0 = pl[first];
    Bins<I> b;
    b.sx = 0; b.sy = 0; b.tx = lut; b.ty = lut;
    bool inside = false;
    int off0 = 0;
    if (ok) {
        b = make_bins<I>(u, v, lut);
        inside = bins_inside<T, 1, I>(b, P0.w, P0.h);
        off0 = b.sy * P0.src_stride + b.sx * (int)sizeof(T);
    }
    const int doff = oy * P0.dst_stride + ox * (int)sizeof(T);
    #pragma unroll 1
    for (int pi = first; pi <= last; ++pi) {
        float o = pl[pi].bg[0];
        if (ok) {
            if (__builtin_expect(inside, 1)) taps_inside<T, 1, I>(pl[pi].src, off0, P0.src_stride, b, pl[pi].limit, &o);
            else taps_edge<T, 1, I>(pl[pi].src, P0.src_stride, b, P0.w, P0.h, pl[pi].bg, pl[pi].limit, &o);
        }
        store_px<T, 1>(pl[pi].dst, doff, &o);
    }
}

// ---- bilinear specialisation (I = 2): named weights, two-compare interior test — the hot configuration ---------------------------------------------------
struct Bins2 { int sx, sy; float cx0, cx1, cy0, cy1; };
__device__ __forceinline__ Bins2 make_bins2(float u, float v) {
    const int sx0 = round_i32(u * 32.0f), sy0 = round_i32(v * 32.0f);
    Bins2 b;
    b.sx = sx0 >> 5; b.sy = sy0 >> 5;
    b.cx1 = (float)(sx0 & 31) * 0.03125f; b.cx0 = 1.0f - b.cx1;     // {1-k/32, k/32}: the LUT row (cpu_undistort.rs:14-19)
    b.cy1 = (float)(sy0 & 31) * 0.03125f; b.cy0 = 1.0f - b.cy1;
    return b;
}
// Taps that straddle the source rect: out-of-rect taps read `bg`, out-of-rect rows contribute bg*cy
// (cpu_undistort.rs:392-409), in the reference's exact operation order.
template <typename T, int N>
__device__ __forceinline__ void taps_edge2(const uint8_t *src, int stride, const Bins2 &b, int w, int h, const float *bg, float limit, float *out) {
    const bool x0in = b.sx >= 0 && b.sx < w, x1in = b.sx + 1 >= 0 && b.sx + 1 < w;
    const bool y0in = b.sy >= 0 && b.sy < h, y1in = b.sy + 1 >= 0 && b.sy + 1 < h;
    const T *row0 = reinterpret_cast<const T *>(src + (int64_t)b.sy * stride) + (int64_t)b.sx * N;
    const T *row1 = reinterpret_cast<const T *>(reinterpret_cast<const uint8_t *>(row0) + stride);
    #pragma unroll
    for (int c = 0; c < N; ++c) {
        const float p00 = (y0in && x0in) ? (float)row0[c] : bg[c];
        const float p01 = (y0in && x1in) ? (float)row0[N + c] : bg[c];
        const float p10 = (y1in && x0in) ? (float)row1[c] : bg[c];
        const float p11 = (y1in && x1in) ? (float)row1[N + c] : bg[c];
        float sum = 0.0f;
        if (y0in) { float xs = 0.0f; xs = xs + p00 * b.cx0; xs = xs + p01 * b.cx1; sum = sum + xs * b.cy0; } else sum = sum + bg[c] * b.cy0;
        if (y1in) { float xs = 0.0f; xs = xs + p10 * b.cx0; xs = xs + p11 * b.cx1; sum = sum + xs * b.cy1; } else sum = sum + bg[c] * b.cy1;
        out[c] = fminf(sum, limit);
    }
}
// All four taps inside.  For the integer pixel types every tap is >= +0, so the reference's leading zero-adds
// (xsum = 0 + p*c, sum = 0 + xs*cy) are exact identities and are dropped; for f32 pixels (-0, negative values) they stay.
template <typename T, int N>
__device__ __forceinline__ void taps_inside2(const uint8_t *src, int off0, int stride, const Bins2 &b, float limit, float *out) {
    const T *row0 = reinterpret_cast<const T *>(src + (int64_t)off0);
    const T *row1 = reinterpret_cast<const T *>(src + (int64_t)(off0 + stride));
    #pragma unroll
    for (int c = 0; c < N; ++c) {
        if (is_f32<T>::value) {
            float xs0 = 0.0f; xs0 = xs0 + (float)row0[c] * b.cx0; xs0 = xs0 + (float)row0[N + c] * b.cx1;
            float xs1 = 0.0f; xs1 = xs1 + (float)row1[c] * b.cx0; xs1 = xs1 + (float)row1[N + c] * b.cx1;
            float sum = 0.0f; sum = sum + xs0 * b.cy0; sum = sum + xs1 * b.cy1;
            out[c] = fminf(sum, limit);
        } else {
            // tap (<= 16 bits) x weight (k/32) and the sum of two such products are exact in f32 (<= 22 bits), so the
            // fused form rounds nowhere the reference's separate multiply and add would
#if GFW_EXP_FMA
            const float xs0 = __builtin_fmaf((float)row0[N + c], b.cx1, (float)row0[c] * b.cx0);
            const float xs1 = __builtin_fmaf((float)row1[N + c], b.cx1, (float)row1[c] * b.cx0);
#else
            const float xs0 = (float)row0[c] * b.cx0 + (float)row0[N + c] * b.cx1;
            const float xs1 = (float)row1[c] * b.cx0 + (float)row1[N + c] * b.cx1;
#endif
#if GFW_EXP_MIN
            out[c] = min_limit(xs0 * b.cy0 + xs1 * b.cy1, limit);
#else
            out[c] = fminf(xs0 * b.cy0 + xs1 * b.cy1, limit);
#endif
        }
    }
}
// One plane.  32-bit byte offsets from the uniform plane base (planes are < 2 GiB, checked on the host).
// Audit mode (aud != nullptr, a compile-time constant after inlining): every byte range about to be touched is checked
// against the length the caller declared for the buffer; violations are counted in aud[5] and the access is skipped.
__device__ __forceinline__ bool range_ok(unsigned long long *aud, int64_t off, int64_t bytes, int len) {
    if (!aud) return true;
    if (off >= 0 && off + bytes <= (int64_t)len) return true;
    atomicAdd(&aud[5], 1ull);
    return false;
}
template <typename T, int N>
__device__ __forceinline__ void sample_store2(float u, float v, bool ok, const GfwYuvPlane &P, const float *bg, float limit, int ox, int oy,
                                              unsigned long long *aud = nullptr) {
    float out[N];
    #pragma unroll
    for (int c = 0; c < N; ++c) out[c] = bg[c];
    if (ok) {
        const Bins2 b = make_bins2(u, v);
        if (__builtin_expect((unsigned)b.sx < (unsigned)(P.w - 1) && (unsigned)b.sy < (unsigned)(P.h - 1), 1)) {
            const int off0 = b.sy * P.src_stride + b.sx * (int)(N * sizeof(T));
            if (range_ok(aud, off0, 2 * N * sizeof(T), P.src_len) && range_ok(aud, (int64_t)off0 + P.src_stride, 2 * N * sizeof(T), P.src_len))
                taps_inside2<T, N>(P.src, off0, P.src_stride, b, limit, out);
        } else
            taps_edge2<T, N>(P.src, P.src_stride, b, P.w, P.h, bg, limit, out);
    }
    const int doff = oy * P.dst_stride + ox * (int)(N * sizeof(T));
    if (range_ok(aud, doff, N * sizeof(T), P.dst_len)) store_px<T, N>(P.dst, doff, out);
}
// Planar planes of identical geometry (U and V; or G,B,R,A of a planar float frame): one set of bins / weights /
// offsets, one gather pair per plane.
template <typename T>
__device__ __forceinline__ void sample_store_shared2(float u, float v, bool ok, const GfwYuvPlane *pl, int first, int last, int ox, int oy) {
    const GfwYuvPlane &P0 = pl[first];
    Bins2 b = {0, 0, 0.0f, 0.0f, 0.0f, 0.0f};
    bool inside = false;
    int off0 = 0;
    if (ok) {
        b = make_bins2(u, v);
        inside = (unsigned)b.sx < (unsigned)(P0.w - 1) && (unsigned)b.sy < (unsigned)(P0.h - 1);
        off0 = b.sy * P0.src_stride + b.sx * (int)sizeof(T);
    }
    const int doff = oy * P0.dst_stride + ox * (int)sizeof(T);
    #pragma unroll 1
    for (int pi = first; pi <= last; ++pi) {
        float o = pl[pi].bg[0];
        if (ok) {
            if (__builtin_expect(inside, 1)) taps_inside2<T, 1>(pl[pi].src, off0, P0.src_stride, b, pl[pi].limit, &o);
            else taps_edge2<T, 1>(pl[pi].src, P0.src_stride, b, P0.w, P0.h, pl[pi].bg, pl[pi].limit, &o);
        }
        store_px<T, 1>(pl[pi].dst, doff, &o);
    }
}

// Two planar chroma planes of identical geometry (U, V) — the C2 hot path: one set of bins / weights / offsets,
// two gathers, no loop over a plane index (which would index the kernel-argument plane array dynamically).
template <typename T>
__device__ __forceinline__ void sample_store_uv2(float u, float v, bool ok, const GfwYuvPlane &PU, const GfwYuvPlane &PV,
                                                 float bg_u, float bg_v, float lim_u, float lim_v, int ox, int oy, unsigned long long *aud = nullptr) {
    float ou = bg_u, ov = bg_v;
    if (ok) {
        const Bins2 b = make_bins2(u, v);
        if (__builtin_expect((unsigned)b.sx < (unsigned)(PU.w - 1) && (unsigned)b.sy < (unsigned)(PU.h - 1), 1)) {
            const int off0 = b.sy * PU.src_stride + b.sx * (int)sizeof(T);
            const int top = PU.src_len < PV.src_len ? PU.src_len : PV.src_len;
            if (range_ok(aud, off0, 2 * sizeof(T), top) && range_ok(aud, (int64_t)off0 + PU.src_stride, 2 * sizeof(T), top)) {
                taps_inside2<T, 1>(PU.src, off0, PU.src_stride, b, lim_u, &ou);
                taps_inside2<T, 1>(PV.src, off0, PU.src_stride, b, lim_v, &ov);
            }
        } else {
            taps_edge2<T, 1>(PU.src, PU.src_stride, b, PU.w, PU.h, &bg_u, lim_u, &ou);
            taps_edge2<T, 1>(PV.src, PU.src_stride, b, PU.w, PU.h, &bg_v, lim_v, &ov);
        }
    }
    const int doff = oy * PU.dst_stride + ox * (int)sizeof(T);
    if (!range_ok(aud, doff, sizeof(T), PU.dst_len < PV.dst_len ? PU.dst_len : PV.dst_len)) return;
    store_px<T, 1>(PU.dst, doff, &ou);
    store_px<T, 1>(PV.dst, doff, &ov);
}

// ---- first pass (rolling-shutter row pick) -----------------------------------------------------------------
// The mid-row projection of undistort_coord (cpu_undistort.rs:470-479) is used for ONE thing: the integer
// sy = clamp(round(p.y)).  FAST1 evaluates p.y with fused arithmetic and a per-lens table of
// s(rho) = theta_d(atan(sqrt(rho)))/sqrt(rho) (linear interpolation, rho = (X/W)^2 + (Y/W)^2) and accepts the
// rounded value only when no half-integer lies within +-E of it, E bounding |approx - exact| (derivation in
// DESIGN.md section 2; tests/test_gpu_pass1.py audits every certificate and measures the real gap).  Everything
// else — a percent or two of the pixels — goes through the exact projection: queued in LDS per wave and resolved
// densely (one exact pass per few rows of the wave instead of one per pixel row).
struct Mid { float m0, m1, m2, m3, m4, m5, m6, m7, m8; };
struct P1 { float rho_max, rho_scale, eps, f, c, lim; };

template <int MODEL>
__device__ __forceinline__ int default_row(float ox, float oy, const GfwYuvArgs &A) {
    const int lim = A.hrs ? A.width : A.height;
    return max(min(round_i32(A.hrs ? ox : oy), lim), 0);
}
// exact: cpu_undistort.rs:465-479
template <int MODEL>
__device__ __forceinline__ int pass1_exact(float ox, float oy, const Mid &M, const Lens &L, const GfwYuvArgs &A) {
    int sy = default_row<MODEL>(ox, oy, A);
    const GfwPt pt = rd<MODEL>(ox, oy, float4{M.m0, M.m1, M.m2, M.m3}, float4{M.m4, M.m5, M.m6, M.m7}, M.m8, A.matrices + (size_t)(A.matrix_count / 2) * GFW_MAT_STRIDE + 8, L, A);
    if (pt.ok) { const int lim = A.hrs ? A.width : A.height; sy = max(min(round_i32(A.hrs ? pt.x : pt.y), lim), 0); }
    return sy;
}
// approximate + certificate; returns false when the exact path must decide.
// (ax, ay, aw) = ox*m0+m2, ox*m3+m5, ox*m6+m8 are per-lane constants of the pixel column.
__device__ __forceinline__ bool pass1_fast(float ax, float ay, float aw, float oy, const Mid &M, const P1 &Q, const float2 *tab,
                                           bool hrs, float rl2, int &sy, float &v_out, unsigned long long *aud = nullptr) {
    const float X = __builtin_fmaf(oy, M.m1, ax);
    const float Y = __builtin_fmaf(oy, M.m4, ay);
    const float W = __builtin_fmaf(oy, M.m7, aw);
    const float rw = gfw_hw_rcp(W);
    const float a = X * rw, b = Y * rw;
    const float rho = __builtin_fmaf(a, a, b * b);
    // W safely positive (the exact path decides validity otherwise) and rho inside the table (NaN fails both)
    bool good = (W > 0.0009765625f) & (rho < Q.rho_max);
    if (rl2 > 0.0f) {                                              // :139 — decide only when clear of the boundary
        const float lhs = __builtin_fmaf(X, X, Y * Y), rhs = rl2 * W;
        good = good & (lhs < rhs * 0.9999f);
    }
    const float tpos = fminf(fmaxf(rho, 0.0f), Q.rho_max) * Q.rho_scale;   // clamped: a rejected lane still indexes the table
    const float ti = floorf(tpos);
    if (aud && !((int)ti >= 0 && (int)ti <= GFW_P1_TABLE_N)) atomicAdd(&aud[5], 1ull);
    const float2 e = tab[(int)ti];
    const float s = __builtin_fmaf(tpos - ti, e.y, e.x);
    const float v = __builtin_fmaf((hrs ? a : b) * s, Q.f, Q.c);
    v_out = v;
    const float g = v - 0.5f;
    const float dist = fabsf(g - rintf(g));                        // distance of v to the nearest half-integer
    const bool outside = !(v > -0.25f) | !(v < Q.lim + 0.25f);     // there the clamp decides and ties cannot matter
    good = good & (outside | (dist > Q.eps)) & (v == v);
    sy = max(min(gfw_f2i(rintf(v)), (int)Q.lim), 0);
    return good;
}

template <int MODEL, typename T, int N0, int I, int DW, int DH, bool INTERLEAVED_UV, int RB, bool FAST1, bool AUDIT>
__global__ __launch_bounds__(256) void gfw_yuv_kernel(const GfwYuvArgs A) {
    // tile = 64 x 4 lanes; each lane owns RB vertically stacked DW x DH luma blocks (+ their chroma sites).
    constexpr int NPX = DW * DH;
    constexpr int QCAP = 128 * NPX;                  // a wave adds at most 64*NPX entries per row; flushed at half full
    static_assert(RB * NPX <= 64, "slot index must fit the 6 low bits of q_dst");
    __shared__ float q_x[FAST1 ? 4 : 1][FAST1 ? QCAP : 1], q_y[FAST1 ? 4 : 1][FAST1 ? QCAP : 1];
    __shared__ unsigned short q_dst[FAST1 ? 4 : 1][FAST1 ? QCAP : 1];           // (owner lane << 6) | slot in s_rows
    __shared__ unsigned q_n[4];
    __shared__ int s_rows[RB * NPX][256];                                        // phase-1 rows, one column per lane
    __shared__ float s_lut[I == 2 ? 1 : 448];                                    // bicubic / Lanczos4 tap table
    const int wave = threadIdx.y, lane = threadIdx.x, tid = wave * 64 + lane;
    if (I != 2) {
        for (int i = tid; i < 448; i += 256) s_lut[i] = GFW_COEFFS[i];
        __syncthreads();
    }
    const bool two_pass = A.matrix_count > 1 && !(A.ablate & 1);
    const bool hrs = A.hrs != 0;

    // uniform floats of the pixel loops, pinned in VGPRs once per wave
    Lens L;
    L.f0 = vu(A.f[0]); L.f1 = vu(A.f[1]); L.c0 = vu(A.c[0]); L.c1 = vu(A.c[1]);
    L.k0 = vu(A.k[0]); L.k1 = vu(A.k[1]); L.k2 = vu(A.k[2]); L.k3 = vu(A.k[3]);
    L.t2x = vu(A.t2[0]); L.t2y = vu(A.t2[1]); L.rl2 = vu(A.r_limit_sq);
    Maps MP;
    MP.mul_lx = vu(A.map_lx.mul); MP.mul_ly = vu(A.map_ly.mul); MP.mul_cx = vu(A.map_cx.mul); MP.mul_cy = vu(A.map_cy.mul);
    MP.den_x = vu(A.map_lx.den); MP.rcp_x = vu(A.map_lx.rcp); MP.den_y = vu(A.map_ly.den); MP.rcp_y = vu(A.map_ly.rcp);
    float bg_y[N0];
    #pragma unroll
    for (int c = 0; c < N0; ++c) bg_y[c] = vu(A.pl[0].bg[c]);
    const float lim_y = vu(A.pl[0].limit);
    float bg_c[2] = {vu(A.pl[1].bg[0]), vu(A.pl[1].bg[1])};
    const float lim_u = vu(A.pl[1].limit), bg_v = vu(A.pl[2].bg[0]), lim_v = vu(A.pl[2].limit);
    Mid M{0, 0, 0, 0, 0, 0, 0, 0, 0};
    P1 Q{0, 0, 0, 0, 0, 0};
    if (two_pass) {
        const float *mid = A.matrices + (size_t)(A.matrix_count >> 1) * GFW_MAT_STRIDE;   // wave-uniform -> scalar loads
        M.m0 = vu(mid[0]); M.m1 = vu(mid[1]); M.m2 = vu(mid[2]); M.m3 = vu(mid[3]); M.m4 = vu(mid[4]);
        M.m5 = vu(mid[5]); M.m6 = vu(mid[6]); M.m7 = vu(mid[7]); M.m8 = vu(mid[8]);
        if (FAST1) {
            Q.rho_max = vu(A.p1_rho_max); Q.rho_scale = vu(A.p1_rho_scale); Q.eps = vu(A.p1_eps);
            Q.f = vu(A.p1_f); Q.c = vu(A.p1_c); Q.lim = vu((float)(A.hrs ? A.width : A.height));
        }
    }

    // persistent walk over this workgroup's share of the XCD band of tiles
    const int n_tiles = A.tiles_x * A.tiles_y;
    const int per_xcd = (n_tiles + 7) >> 3;
    const int wg_per_xcd = (int)gridDim.x >> 3;
    const int xcd = (int)blockIdx.x & 7;
    for (int tb = (int)blockIdx.x >> 3; tb < per_xcd; tb += wg_per_xcd) {
        const int t = xcd * per_xcd + tb;
        if (t >= n_tiles) break;
        const int ty = t / A.tiles_x, tx = t - ty * A.tiles_x;
        const int cx = tx * 64 + lane;
        const int cy0 = (ty * 4 + wave) * RB;            // first chroma-site row of this lane
        const bool lane_ok = cx < A.cw;

        // ---- phase 1: rolling-shutter row of every luma pixel of this lane ----------------------------
        if (two_pass) {
            if (FAST1) {
                if (lane == 0) q_n[wave] = 0;
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            }
            #pragma unroll 1
            for (int r = 0; r < RB; ++r) {
                #pragma unroll (NPX <= 2 ? NPX : 1)
                for (int k = 0; k < NPX; ++k) {
                    const int i = k % DW, j = k / DW;
                    const int lx = cx * DW + i, ly = (cy0 + r) * DH + j;
                    int sy = 0;
                    if (lane_ok && lx < A.out_w && ly < A.out_h) {
                        const float ox = (float)lx + L.t2x, oy = (float)ly + L.t2y;
                        if (FAST1) {
                            float v_fast;
                            const float ax = __builtin_fmaf(ox, M.m0, M.m2), ay = __builtin_fmaf(ox, M.m3, M.m5), aw = __builtin_fmaf(ox, M.m6, M.m8);
                            if (!pass1_fast(ax, ay, aw, oy, M, Q, A.p1_table, hrs, L.rl2, sy, v_fast, AUDIT ? A.audit : nullptr)) {
                                const unsigned slot = atomicAdd(&q_n[wave], 1u);      // < QCAP: flushed below before it can fill
                                q_x[wave][slot] = ox; q_y[wave][slot] = oy;
                                q_dst[wave][slot] = (unsigned short)((lane << 6) | (r * NPX + k));
                                if (AUDIT) atomicAdd(&A.audit[2], 1ull);
                            } else if (AUDIT) {                                       // audit: every certificate is checked
                                atomicAdd(&A.audit[0], 1ull);
                                if (pass1_exact<MODEL>(ox, oy, M, L, A) != sy) atomicAdd(&A.audit[1], 1ull);
                                const GfwPt ex = rd<MODEL>(ox, oy, float4{M.m0, M.m1, M.m2, M.m3}, float4{M.m4, M.m5, M.m6, M.m7}, M.m8, A.matrices + (size_t)(A.matrix_count / 2) * GFW_MAT_STRIDE + 8, L, A);
                                if (ex.ok) atomicMax(&A.audit[4], (unsigned long long)gfw_f2u(fabsf((hrs ? ex.x : ex.y) - v_fast)));
                            }
                        } else {
                            sy = pass1_exact<MODEL>(ox, oy, M, L, A);
                        }
                    }
                    s_rows[r * NPX + k][tid] = sy;
                }
                if (FAST1) {
                    // ---- phase 2: the wave resolves its queued pixels exactly, densely packed.  Flushed after
                    // the last row, or earlier when the next row (<= 64*NPX new entries) could overflow the queue.
                    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                    const unsigned qn = q_n[wave];
                    if (r == RB - 1 || qn + 64u * NPX > (unsigned)QCAP) {
                        for (unsigned e = lane; e < qn; e += 64) {
                            const int sy = pass1_exact<MODEL>(q_x[wave][e], q_y[wave][e], M, L, A);
                            const unsigned d = q_dst[wave][e];
                            s_rows[d & 63u][wave * 64 + (d >> 6)] = sy;
                        }
                        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                        if (lane == 0) q_n[wave] = 0;
                        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                    }
                }
            }
        }

        // ---- phase 3: exact projection with the row's own matrix, then taps ---------------------------
        if (lane_ok) {
            #pragma unroll 1
            for (int r = 0; r < RB; ++r) {
                const int cy = cy0 + r;
                if (cy >= A.ch) break;
                float u0 = 0.0f, v0 = 0.0f; bool ok0 = false;
                #pragma unroll (NPX <= 2 ? NPX : 1)
                for (int k = 0; k < NPX; ++k) {
                    const int i = k % DW, j = k / DW;
                    const int lx = cx * DW + i, ly = cy * DH + j;
                    if (lx >= A.out_w || ly >= A.out_h) continue;
                    const float ox = (float)lx + L.t2x, oy = (float)ly + L.t2y;
                    const int sy = two_pass ? s_rows[r * NPX + k][tid] : default_row<MODEL>(ox, oy, A);
                    GfwPt p;
                    if (A.ablate & 8) { p.x = ox * 0.5f; p.y = oy * 0.5f; p.ok = true; }              // timing ablation only
                    else {
                        const int row = min(sy, A.matrix_count - 1);
                        if (AUDIT && (unsigned)row >= (unsigned)A.matrix_count) atomicAdd(&A.audit[5], 1ull);
                        p = rd_row<MODEL>(ox, oy, row, L, A);
                    }
                    if (A.background_mode != 0 && p.ok) {                      // cpu_undistort.rs:495-509 (edge repeat / edge mirror)
                        const float width_f = (float)A.width, height_f = (float)A.height;
                        if (A.background_mode == 1) {
                            p.x = fminf(fmaxf(p.x, 3.0f), width_f - 3.0f);
                            p.y = fminf(fmaxf(p.y, 3.0f), height_f - 3.0f);
                        } else {
                            const float rx = roundf(p.x), ry = roundf(p.y);
                            const float width3 = width_f - 3.0f, height3 = height_f - 3.0f;
                            if (rx > width3)  p.x = width3  - (rx - width3);
                            if (rx < 3.0f)    p.x = 3.0f + width_f - (width3  + rx);
                            if (ry > height3) p.y = height3 - (ry - height3);
                            if (ry < 3.0f)    p.y = 3.0f + height_f - (height3 + ry);
                        }
                    }
                    if (k == 0) { u0 = p.x; v0 = p.y; ok0 = p.ok; }
                    const float lu = map_c<MODEL != GFW_MODEL_OPENCV_FISHEYE>(p.x, MP.mul_lx, MP.den_x, MP.rcp_x), lv = map_c<MODEL != GFW_MODEL_OPENCV_FISHEYE>(p.y, MP.mul_ly, MP.den_y, MP.rcp_y);   // cpu_undistort.rs:511-514
                    if (A.ablate & 2) { if (lane == 99) A.pl[0].dst[0] = (uint8_t)(lu + lv); continue; }  // timing ablation only
                    if (I == 2) sample_store2<T, N0>(lu, lv, p.ok, A.pl[0], bg_y, lim_y, lx, ly, AUDIT ? A.audit : nullptr);
                    else sample_store<T, N0, I>(lu, lv, p.ok, A.pl[0], bg_y, lim_y, lx, ly, s_lut);
                }
                if (A.nplanes > 1 && !(A.ablate & 4)) {
                    const float cu = map_c<MODEL != GFW_MODEL_OPENCV_FISHEYE>(u0, MP.mul_cx, MP.den_x, MP.rcp_x), cv = map_c<MODEL != GFW_MODEL_OPENCV_FISHEYE>(v0, MP.mul_cy, MP.den_y, MP.rcp_y);
                    if (I == 2) {
                        if (INTERLEAVED_UV) sample_store2<T, 2>(cu, cv, ok0, A.pl[1], bg_c, lim_u, cx, cy, AUDIT ? A.audit : nullptr);
                        else if (A.nplanes == 3) sample_store_uv2<T>(cu, cv, ok0, A.pl[1], A.pl[2], bg_c[0], bg_v, lim_u, lim_v, cx, cy, AUDIT ? A.audit : nullptr);
                        else sample_store_shared2<T>(cu, cv, ok0, A.pl, 1, A.nplanes - 1, cx, cy);
                    } else {
                        if (INTERLEAVED_UV) sample_store<T, 2, I>(cu, cv, ok0, A.pl[1], bg_c, lim_u, cx, cy, s_lut);
                        else sample_store_shared<T, I>(cu, cv, ok0, A.pl, 1, A.nplanes - 1, cx, cy, s_lut);
                    }
                }
            }
        }
        if (FAST1 && two_pass) __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");   // s_rows is rewritten by the next tile
    }
}

template <int MODEL, typename T, int N0, int I, int RB, bool FAST1, bool AUDIT>
hipError_t launch_mt(const GfwYuvArgs &A, int dw, int dh, bool interleaved, hipStream_t s) {
    const int n_tiles = A.tiles_x * A.tiles_y;
    if (n_tiles <= 0) return hipSuccess;
    // persistent grid: a multiple of 8 (XCD bands), ~6 workgroups per CU by default, never more than one per tile
    int grid = A.grid_limit > 0 ? A.grid_limit : 256 * 6;
    const int per_xcd = (n_tiles + 7) >> 3;
    if (grid > per_xcd * 8) grid = per_xcd * 8;
    grid = (grid + 7) & ~7;
    dim3 block(64, 4);
#define GFW_YUV_LAUNCH(DW, DH, IL) hipLaunchKernelGGL((gfw_yuv_kernel<MODEL, T, N0, I, DW, DH, IL, RB, FAST1, AUDIT>), dim3(grid), block, 0, s, A)
    if (N0 > 1 || is_f32<T>::value) {                    // packed single plane, or planar f32 planes: full resolution only
        if (dw == 1 && dh == 1 && !interleaved) GFW_YUV_LAUNCH(1, 1, false);
        else return hipErrorInvalidValue;
    } else {
        if (dw == 2 && dh == 1 && !interleaved) GFW_YUV_LAUNCH(2, 1, false);
        else if (dw == 2 && dh == 1 && interleaved) GFW_YUV_LAUNCH(2, 1, true);
        else if (dw == 2 && dh == 2 && !interleaved) GFW_YUV_LAUNCH(2, 2, false);
        else if (dw == 2 && dh == 2 && interleaved) GFW_YUV_LAUNCH(2, 2, true);
        else if (dw == 1 && dh == 1 && !interleaved) GFW_YUV_LAUNCH(1, 1, false);
        else if (dw == 1 && dh == 1 && interleaved) GFW_YUV_LAUNCH(1, 1, true);
        else return hipErrorInvalidValue;
    }
#undef GFW_YUV_LAUNCH
    return hipGetLastError();
}

}  // namespace

template <int MODEL, typename T, int N0>
static hipError_t launch_tn(const GfwYuvArgs &A, int dw, int dh, bool interleaved, bool fast1, hipStream_t s) {
    constexpr int I = GFW_FRAME_TAPS;
    if (MODEL == GFW_MODEL_OPENCV_FISHEYE && fast1) {
        if (I == 2 && A.audit) return launch_mt<MODEL, T, N0, I, GFW_YUV_RB_FAST, true, (I == 2)>(A, dw, dh, interleaved, s);
        return launch_mt<MODEL, T, N0, I, GFW_YUV_RB_FAST, true, false>(A, dw, dh, interleaved, s);
    }
    return launch_mt<MODEL, T, N0, I, GFW_YUV_RB_EXACT, false, false>(A, dw, dh, interleaved, s);
}
// This translation unit is compiled once per (sample kind, tap count): -DGFW_FRAME_KIND=1|2|4 (u8, u16, f32) and
// -DGFW_FRAME_TAPS=2|4|8 (bilinear, bicubic, Lanczos4), so that the nine families of instantiations build in parallel;
// gfw_kernels.hip dispatches on both.
#if !defined(GFW_FRAME_KIND) || !defined(GFW_FRAME_TAPS)
#error "compile with -DGFW_FRAME_KIND=1|2|4 -DGFW_FRAME_TAPS=2|4|8"
#endif
template <int MODEL>
static hipError_t launch_m(const GfwYuvArgs &A, int n0, int dw, int dh, bool interleaved, bool fast1, hipStream_t s) {
#if GFW_FRAME_KIND == 1
    if (n0 == 1) return launch_tn<MODEL, uint8_t, 1>(A, dw, dh, interleaved, fast1, s);
    if (n0 == 3) return launch_tn<MODEL, uint8_t, 3>(A, dw, dh, interleaved, fast1, s);
    if (n0 == 4) return launch_tn<MODEL, uint8_t, 4>(A, dw, dh, interleaved, fast1, s);
#elif GFW_FRAME_KIND == 2
    if (n0 == 1) return launch_tn<MODEL, uint16_t, 1>(A, dw, dh, interleaved, fast1, s);
    if (n0 == 3) return launch_tn<MODEL, uint16_t, 3>(A, dw, dh, interleaved, fast1, s);
    if (n0 == 4) return launch_tn<MODEL, uint16_t, 4>(A, dw, dh, interleaved, fast1, s);
#else
    if (n0 == 1) return launch_tn<MODEL, float, 1>(A, dw, dh, interleaved, fast1, s);
    if (n0 == 4) return launch_tn<MODEL, float, 4>(A, dw, dh, interleaved, fast1, s);
#endif
    return hipErrorInvalidValue;
}

#define GFW_CAT2(a, b) a##b
#define GFW_CAT(a, b) GFW_CAT2(a, b)
#define GFW_FN GFW_CAT(GFW_CAT(gfw_launch_yuv_kind, GFW_FRAME_KIND), GFW_CAT(_taps, GFW_FRAME_TAPS))
hipError_t GFW_FN(const GfwYuvArgs &A, int n0, int dw, int dh, bool interleaved, bool fast1, hipStream_t s) {
    if (A.model == GFW_MODEL_OPENCV_FISHEYE && !A.extras) return launch_m<GFW_MODEL_OPENCV_FISHEYE>(A, n0, dw, dh, interleaved, fast1, s);
    return launch_m<-1>(A, n0, dw, dh, interleaved, false, s);
}
