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
#include <cstdio>
#include <type_traits>
#include <cstdlib>

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
#ifndef GFW_DOT_TAPS_U16
#define GFW_DOT_TAPS_U16 0       // integer-dot taps for 16-bit planes in the default kernel: measured SLOWER on MI355X (96.5 vs 81.7 us per 4K
                                 // frame: two dword gathers at 2-byte alignment against four aligned 16-bit ones); 8-bit planes gain (C1: 18.0 -> 16.3 us)
#endif
#ifndef GFW_WAVES_PER_EU
#define GFW_WAVES_PER_EU 6       // register budget of the frame kernels, in waves per SIMD (512 / N VGPRs).  6: 71 VGPRs, 106 SGPRs = six workgroups
                                 // per CU.  7 (94 SGPRs, more scalar reloads) measures 2-4 % slower even with seven workgroups per CU resident, 8
                                 // (78 SGPRs: 700 v_readlane) 6 % slower: profiles/r02_scheduling_experiments.md
#endif
#ifndef GFW_PASS1_PAIR
#define GFW_PASS1_PAIR 0         // first pass of the lane's pixel pair with packed math (hot_pass1_pair) in the default kernel: measured 11 % slower
#endif
#ifndef GFW_HOT_ONLY
#define GFW_HOT_ONLY 0           // A/B builds (tools/build_variants.sh): only the C2 instantiation (u16, 4:2:2 planar, bilinear), seconds to compile
#endif
#ifndef GFW_XCD_CHUNK
#define GFW_XCD_CHUNK 0          // 0: each XCD walks one contiguous band of tiles; C > 0: chunks of C consecutive tiles are dealt round-robin to the
                                 // XCDs (the bands differ by 3 % in cost; measured +0..3 %, inside the run-to-run noise: not enabled)
#endif
#ifndef GFW_LDS_MATRICES
#define GFW_LDS_MATRICES 0       // second pass reads its matrix rows from a per-wave LDS window (32 rows from the tile's smallest row index) instead
                                 // of L1/L2 — the north-star's "per-row matrices staged in LDS".  Certified-first-pass kernels only.  Staged for A/B
#endif
#ifndef GFW_LUT_TILE
#define GFW_LUT_TILE 0           // bicubic / Lanczos4 taps of planar 8/16-bit frames from a per-wave LDS tile of the source (tile_sample_store):
                                 // one coalesced fetch of the wave's bounding box per output row and plane instead of I row fetches per sample.
                                 // Written, compiles, not yet run on the device: off in the shipped binary
#endif
#ifndef GFW_STAGED_FUSED
#define GFW_STAGED_FUSED 0       // 1: build the fused paths that are written but not yet through the GPU parity suite (background mode 3, Sony
                                 // mesh; tests/test_staged_fused_coverage.py, GFW_OPT_KERNEL_VARIANT = 7).  0: they do not exist in the binary —
                                 // inside the generic-model instantiation they cost every other user of it registers and scratch
#endif
#ifndef GFW_GENERIC_WAVES_PER_EU
#define GFW_GENERIC_WAVES_PER_EU 3   // register budget of the generic-model instantiations (see the kernel's attribute)
#endif
#ifndef GFW_ATAN_TABLE
#define GFW_ATAN_TABLE 0         // exact projection's atanf with the table-driven reduction (gfw_fastmath.h: gfw_atanf_pos_tab): bit-identical on
                                 // the host, ~17 instructions fewer per projection; not yet timed on the device
#endif
#ifndef GFW_PRIO_MODE
#define GFW_PRIO_MODE 1          // wave issue priority by remaining work (s_setprio).  The SIMD arbiter serves the oldest wave first, so the six
                                 // waves of a SIMD progress at 0.115 ... 0.196 lane-rows/us and finish up to 17 us apart
                                 // (profiles/r02_wave_timeline.txt).  1: priority = min(3, remaining lane-rows / GFW_PRIO_DIV), re-evaluated every
                                 // row: waves with more work left are served first and the finish times close up — C2: 80.5 -> 76.4 us per frame
                                 // (DIV 3; 78.1 with 2 or 4).  2: min(3, remaining tiles): no gain.  0: off.
#endif
#ifndef GFW_PRIO_DIV
#define GFW_PRIO_DIV 3
#endif
#ifndef GFW_PRIO_AGE_ROWS
#define GFW_PRIO_AGE_ROWS 0      // A/B: lane-rows of head start the LAST workgroup dispatched to a CU gets in the priority formula (earlier ones
                                 // proportionally less): counters the arbiter's oldest-first rule directly.  0 = off
#endif
#ifndef GFW_TIMELINE
#define GFW_TIMELINE 0           // diagnosis builds only: per-wave start / end / phase clocks and HW_ID into a device array that the 60th launch
                                 // dumps to $GFW_TIMELINE_FILE (tools/analyze_timeline.py)
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
#ifndef GFW_PIN_LENS
#define GFW_PIN_LENS 0            // only the eight lens constants of the exact projection (f, c, k0..k3) pinned in VGPRs: the compiler otherwise
                                  // re-reads them from the kernel-argument segment inside the pixel loop (s_load_dwordx8 + wait, twice per pixel)
#endif
__device__ __forceinline__ float vu_lens(float s) {
#if GFW_PIN_LENS
    float v; asm volatile("v_mov_b32 %0, %1" : "=v"(v) : "s"(s)); return v;
#else
    return s;
#endif
}
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
#if GFW_ATAN_TABLE
        return gfw_atanf_pos_tab(x);
#else
        return gfw_atanf_pos(x);
#endif
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
#if GFW_STAGED_FUSED
        if (A.extras & 32) gfw_mesh_apply(u, v, A.kp, A.common);                   // Sony mesh + focal-plane distortion (:169-214)
#endif
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

// ---- integer-dot taps for 8/16-bit planes ---------------------------------------------------------------------------------
// A bilinear sample of an integer plane is sum = RN(RN(xs0*cy0) + RN(xs1*cy1)) with xs = p0*(1-k/32) + p1*k/32 exact
// (cpu_undistort.rs:392-411).  xs*32 = p0*(32-k) + p1*k is ONE integer dot instruction on the raw loaded word
// (v_dot2_u32_u16 / v_dot4_u32_u8), converted exactly (< 2^22), multiplied by the integer y weight and scaled by 2^-10 at the
// end — power-of-two scaling commutes with round-to-nearest, so the two roundings are the reference's.
typedef float gfw_f2 __attribute__((ext_vector_type(2)));
typedef unsigned short gfw_us2 __attribute__((ext_vector_type(2)));

template <typename T, bool UV> struct HotTap;
template <> struct HotTap<uint16_t, false> {                 // two u16 taps: two ALIGNED 16-bit gathers packed into one word (a dword gather at
    static constexpr int BYTES = 4, PX = 2;                  // 2-byte alignment costs the texture-address path ~3x: profiles/r01_membench_*.txt)
    static __device__ __forceinline__ uint32_t load(const uint8_t *src, uint32_t off) {
        const uint32_t lo = *reinterpret_cast<const uint16_t *>(src + off), hi = *reinterpret_cast<const uint16_t *>(src + off + 2u);
        return lo | (hi << 16);
    }
    static __device__ __forceinline__ uint32_t wpack(uint32_t k) { return (32u - k) | (k << 16); }
    static __device__ __forceinline__ uint32_t dot(uint32_t raw, uint32_t w) { return __builtin_amdgcn_udot2(__builtin_bit_cast(gfw_us2, raw), __builtin_bit_cast(gfw_us2, w), 0u, false); }
};
template <> struct HotTap<uint8_t, false> {                  // two u8 taps = one 16-bit word at any address
    static constexpr int BYTES = 2, PX = 1;
    typedef uint16_t u16u __attribute__((aligned(1)));
    static __device__ __forceinline__ uint32_t load(const uint8_t *src, uint32_t off) { return *reinterpret_cast<const u16u *>(src + off); }
    static __device__ __forceinline__ uint32_t wpack(uint32_t k) { return (32u - k) | (k << 8); }
    static __device__ __forceinline__ uint32_t dot(uint32_t raw, uint32_t w) { return __builtin_amdgcn_udot4(raw, w, 0u, false); }
};
// interleaved chroma: (U0 V0 U1 V1)
template <> struct HotTap<uint8_t, true> {                   // four bytes at a 2-byte aligned address
    static constexpr int BYTES = 4, PX = 2;
    typedef uint32_t u32u __attribute__((aligned(2)));
    static __device__ __forceinline__ uint32_t load(const uint8_t *src, uint32_t off) { return *reinterpret_cast<const u32u *>(src + off); }
    static __device__ __forceinline__ uint32_t wpack(uint32_t k) { return (32u - k) | (k << 16); }          // bytes 0 and 2 (U); << 8 for V
    static __device__ __forceinline__ void dot(uint32_t raw, uint32_t w, uint32_t &u, uint32_t &v) {
        u = __builtin_amdgcn_udot4(raw, w, 0u, false); v = __builtin_amdgcn_udot4(raw, w << 8, 0u, false);
    }
};
template <> struct HotTap<uint16_t, true> {                  // eight bytes = two aligned dwords (a UV16 pixel is 4 bytes)
    static constexpr int BYTES = 8, PX = 4;
    static __device__ __forceinline__ uint2 load(const uint8_t *src, uint32_t off) {
        typedef uint32_t u32u __attribute__((aligned(2)));
        return uint2{*reinterpret_cast<const u32u *>(src + off), *reinterpret_cast<const u32u *>(src + off + 4u)};
    }
    static __device__ __forceinline__ uint32_t wpack(uint32_t k) { return (32u - k) | (k << 16); }
    static __device__ __forceinline__ void dot(uint2 raw, uint32_t w, uint32_t &u, uint32_t &v) {
        const uint32_t uu = __builtin_amdgcn_perm(raw.y, raw.x, 0x05040100u);      // (U0, U1)
        const uint32_t vv = __builtin_amdgcn_perm(raw.y, raw.x, 0x07060302u);      // (V0, V1)
        u = __builtin_amdgcn_udot2(__builtin_bit_cast(gfw_us2, uu), __builtin_bit_cast(gfw_us2, w), 0u, false);
        v = __builtin_amdgcn_udot2(__builtin_bit_cast(gfw_us2, vv), __builtin_bit_cast(gfw_us2, w), 0u, false);
    }
};
__device__ __forceinline__ uint32_t hot_f2u(float v) { uint32_t r; asm("v_cvt_u32_f32 %0, %1" : "=v"(r) : "v"(v)); return r; }
// RN(RN(xs0*cy0) + RN(xs1*cy1)) from the two integer row sums (see the header comment), clamped by pixel_value_limit
__device__ __forceinline__ uint32_t hot_blend(uint32_t i0, uint32_t i1, uint32_t ky, float limit) {
    const float s = ((float)i0 * (float)(32u - ky) + (float)i1 * (float)ky) * 0.0009765625f;
    return hot_f2u(min_limit(s, limit));
}

// ---- bilinear specialisation (I = 2): named weights, two-compare interior test — the hot configuration ---------------------------------------------------
struct Bins2 { int sx, sy; float cx0, cx1, cy0, cy1; uint32_t kx, ky; };
__device__ __forceinline__ Bins2 make_bins2(float u, float v) {
    const int sx0 = round_i32(u * 32.0f), sy0 = round_i32(v * 32.0f);
    Bins2 b;
    b.sx = sx0 >> 5; b.sy = sy0 >> 5;
    b.kx = (uint32_t)sx0 & 31u; b.ky = (uint32_t)sy0 & 31u;
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
            if (range_ok(aud, off0, 2 * N * sizeof(T), P.src_len) && range_ok(aud, (int64_t)off0 + P.src_stride, 2 * N * sizeof(T), P.src_len)) {
                if constexpr (!is_f32<T>::value && (N == 1 || N == 2) && (sizeof(T) == 1 || GFW_DOT_TAPS_U16)) {
                    // integer-dot taps: the pixel value comes out as an integer; store it and leave
                    const uint32_t doff = (uint32_t)oy * (uint32_t)P.dst_stride + (uint32_t)ox * (uint32_t)(N * sizeof(T));
                    if (!range_ok(aud, doff, N * sizeof(T), P.dst_len)) return;
                    if constexpr (N == 1) {
                        typedef HotTap<T, false> Tap;
                        const uint32_t r0 = Tap::load(P.src, (uint32_t)off0), r1 = Tap::load(P.src, (uint32_t)off0 + (uint32_t)P.src_stride);
                        const uint32_t w = Tap::wpack(b.kx);
                        *reinterpret_cast<T *>(P.dst + doff) = (T)hot_blend(Tap::dot(r0, w), Tap::dot(r1, w), b.ky, limit);
                    } else {
                        typedef HotTap<T, true> Tap;
                        const auto r0 = Tap::load(P.src, (uint32_t)off0), r1 = Tap::load(P.src, (uint32_t)off0 + (uint32_t)P.src_stride);
                        const uint32_t w = Tap::wpack(b.kx);
                        uint32_t u0, v0, u1, v1;
                        Tap::dot(r0, w, u0, v0); Tap::dot(r1, w, u1, v1);
                        const uint32_t ou = hot_blend(u0, u1, b.ky, limit), ov = hot_blend(v0, v1, b.ky, limit);
                        T *d = reinterpret_cast<T *>(P.dst + doff);
                        d[0] = (T)ou; d[1] = (T)ov;
                    }
                    return;
                } else {
                    taps_inside2<T, N>(P.src, off0, P.src_stride, b, limit, out);
                }
            }
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
    Bins2 b = {0, 0, 0.0f, 0.0f, 0.0f, 0.0f, 0u, 0u};
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
                if constexpr (!is_f32<T>::value && (sizeof(T) == 1 || GFW_DOT_TAPS_U16)) {
                    typedef HotTap<T, false> Tap;
                    const uint32_t doff = (uint32_t)oy * (uint32_t)PU.dst_stride + (uint32_t)ox * (uint32_t)sizeof(T);
                    if (!range_ok(aud, doff, sizeof(T), PU.dst_len < PV.dst_len ? PU.dst_len : PV.dst_len)) return;
                    const uint32_t a0 = Tap::load(PU.src, (uint32_t)off0), a1 = Tap::load(PU.src, (uint32_t)off0 + (uint32_t)PU.src_stride);
                    const uint32_t b0 = Tap::load(PV.src, (uint32_t)off0), b1 = Tap::load(PV.src, (uint32_t)off0 + (uint32_t)PU.src_stride);
                    const uint32_t w = Tap::wpack(b.kx);
                    *reinterpret_cast<T *>(PU.dst + doff) = (T)hot_blend(Tap::dot(a0, w), Tap::dot(a1, w), b.ky, lim_u);
                    *reinterpret_cast<T *>(PV.dst + doff) = (T)hot_blend(Tap::dot(b0, w), Tap::dot(b1, w), b.ky, lim_v);
                    return;
                } else {
                    taps_inside2<T, 1>(PU.src, off0, PU.src_stride, b, lim_u, &ou);
                    taps_inside2<T, 1>(PV.src, off0, PU.src_stride, b, lim_v, &ov);
                }
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

#if GFW_STAGED_FUSED
// ---- background mode 3 ("margin with feather", cpu_undistort.rs:576-613) ----------------------------------------------------
// Near the frame border the pixel is c1 * alpha + c2 * (1 - alpha): c1 sampled at the projected point, c2 at the point pulled
// towards the centre by background_margin, alpha the distance to the border in units of the feather.  uv lives in full-resolution
// coordinates for every plane, so alpha and the second point are the same for a luma pixel and the chroma site that shares its
// coordinate.  Served by the generic-model instantiation only (extras & 16).
struct Feather { float alpha, x2, y2; };
__device__ __forceinline__ Feather feather_of(float ux, float uy, const GfwYuvArgs &A) {
    const float width_f = (float)A.width, height_f = (float)A.height;
    const float widthf = width_f - 1.0f, heightf = height_f - 1.0f;
    const float feather = fmaxf(A.kp.background_margin_feather * heightf, 0.0001f);
    Feather f{1.0f, ux, uy};
    if ((ux > widthf - feather) || (ux < feather) || (uy > heightf - feather) || (uy < feather)) {
        f.alpha = fmaxf(fminf(fminf(fminf(fminf(widthf - ux, heightf - uy), ux), uy) / feather, 1.0f), 0.0f);
        float p2x = ux / width_f, p2y = uy / height_f;
        p2x = ((p2x - 0.5f) * (1.0f - A.kp.background_margin)) + 0.5f;
        p2y = ((p2y - 0.5f) * (1.0f - A.kp.background_margin)) + 0.5f;
        f.x2 = p2x * width_f; f.y2 = p2y * height_f;
    }
    return f;
}
// sample_input_at for the LUT samplers (cpu_undistort.rs:371-418) without the store: N channels of one plane at (u, v).
template <typename T, int N, int I>
__device__ __forceinline__ void sample_only(float u, float v, const GfwYuvPlane &P, const float *bg, float limit, const float *lut, float *out) {
    if (I == 2) {
        const Bins2 b = make_bins2(u, v);
        if ((unsigned)b.sx < (unsigned)(P.w - 1) && (unsigned)b.sy < (unsigned)(P.h - 1))
            taps_inside2<T, N>(P.src, b.sy * P.src_stride + b.sx * (int)(N * sizeof(T)), P.src_stride, b, limit, out);
        else
            taps_edge2<T, N>(P.src, P.src_stride, b, P.w, P.h, bg, limit, out);
    } else {
        const Bins<I> b = make_bins<I>(u, v, lut);
        if (bins_inside<T, N, I>(b, P.w, P.h))
            taps_inside<T, N, I>(P.src, b.sy * P.src_stride + b.sx * (int)(N * sizeof(T)), P.src_stride, b, limit, out);
        else
            taps_edge<T, N, I>(P.src, P.src_stride, b, P.w, P.h, bg, limit, out);
    }
}
// One plane's pixel in background mode 3: two samples, blended, stored.  (mul_x, mul_y) = the plane's source_rect map.
template <typename T, int N, int I, bool INF_SAFE>
__device__ __forceinline__ void feather_store(float ux, float uy, const Feather &f, const GfwYuvPlane &P, const float *bg, float limit,
                                              float mul_x, float mul_y, const Maps &MP, int ox, int oy, const float *lut) {
    float c1[N], c2[N], px[N];
    sample_only<T, N, I>(map_c<INF_SAFE>(ux, mul_x, MP.den_x, MP.rcp_x), map_c<INF_SAFE>(uy, mul_y, MP.den_y, MP.rcp_y), P, bg, limit, lut, c1);
    sample_only<T, N, I>(map_c<INF_SAFE>(f.x2, mul_x, MP.den_x, MP.rcp_x), map_c<INF_SAFE>(f.y2, mul_y, MP.den_y, MP.rcp_y), P, bg, limit, lut, c2);
    #pragma unroll
    for (int c = 0; c < N; ++c) px[c] = c1[c] * f.alpha + c2[c] * (1.0f - f.alpha);
    store_px<T, N>(P.dst, oy * P.dst_stride + ox * (int)(N * sizeof(T)), px);
}

#endif   // GFW_STAGED_FUSED

#if GFW_LUT_TILE
// ---- LUT taps from an LDS tile (bicubic / Lanczos4, single-channel 8/16-bit planes) -------------------------------------------
// The I x I windows of a wave's samples of one output row overlap almost entirely (neighbouring lanes are ~1 source pixel apart),
// yet taps_inside fetches I rows per sample, one waited-for row at a time.  Here the wave copies the bounding box of its windows —
// GFW_TILE_H rows of GFW_TILE_W source pixels starting at the wave-wide minimum (sx, sy) — into LDS with one 8-byte fetch per lane
// and row, all in flight together, and every sample whose window lies inside that box takes its taps from LDS in the reference's
// order (cpu_undistort.rs:391-411: xs = xs + p*cx over a row, sum = sum + xs*cy over the rows).  Samples that do not fit (steep
// rotation, zoom-out beyond 160/128, frame edges) go through sample_store as before.  Needs every lane of the wave active (the
// copy is cooperative), the plane 4-byte aligned with a 4-byte multiple pitch.
constexpr int GFW_TILE_W = 160, GFW_TILE_H = 12;
template <typename T, int I, int NS>
__device__ __forceinline__ void tile_sample_store(const GfwYuvPlane &P, const float *u, const float *v, const bool *ok, const bool *need,
                                                  const int *ox, const int *oy, const float *bg, float limit,
                                                  uint2 *tile, int *org, const float *lut, int lane) {
    // GFW_LUT_TILE = 3: the tile holds f32 (each source pixel converted once, two instead of three instructions per tap, twice the
    // LDS: three workgroups per CU); otherwise the raw 8/16-bit pixels
    typedef typename std::conditional<GFW_LUT_TILE == 3, float, T>::type E;
    constexpr int TWB = GFW_TILE_W * (int)sizeof(T);           // source bytes per tile row
    constexpr int CH = TWB / 8;                                // 8-byte source chunks per row = lanes that copy
    constexpr int PXC = 8 / (int)sizeof(T);                    // pixels per chunk
    static_assert(TWB % 8 == 0 && CH <= 64, "tile row must be whole 8-byte chunks, one per lane");
    Bins<I> b[NS];
    bool inside[NS];
    #pragma unroll
    for (int q = 0; q < NS; ++q) {
        inside[q] = false;
        b[q].sx = 0; b[q].sy = 0; b[q].tx = lut; b[q].ty = lut;
        if (need[q] && ok[q]) { b[q] = make_bins<I>(u[q], v[q], lut); inside[q] = bins_inside<T, 1, I>(b[q], P.w, P.h); }
    }
    const bool all_lanes = __builtin_amdgcn_read_exec() == ~0ull;
    const bool aligned = ((P.src_stride & 3) == 0) && (((uintptr_t)P.src & 3u) == 0);
    bool have = false;
    int x0 = 0, y0 = 0, valid_w = 0;
    if (all_lanes && aligned) {
        if (lane == 0) { org[0] = 0x7fffffff; org[1] = 0x7fffffff; }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        #pragma unroll
        for (int q = 0; q < NS; ++q) if (inside[q]) { atomicMin(&org[0], b[q].sx); atomicMin(&org[1], b[q].sy); }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        const int mx = __builtin_amdgcn_readfirstlane(org[0]);
        y0 = __builtin_amdgcn_readfirstlane(org[1]);
        if (mx != 0x7fffffff) {
            have = true;
            x0 = mx & ~(int)(4 / (int)sizeof(T) - 1);                      // the box starts on a 4-byte boundary of the row
            const int row_bytes = P.w * (int)sizeof(T), x0b = x0 * (int)sizeof(T);
            const int chunks = min(CH, (row_bytes - x0b) / 8);             // whole chunks that lie inside the row
            valid_w = chunks * (8 / (int)sizeof(T));
            const int xb = x0b + 8 * lane;
            #pragma unroll 1
            for (int r = 0; r < GFW_TILE_H; ++r) {
                const int y = y0 + r;
                if (y >= P.h) break;                                       // rows past the plane are never part of an inside window
                if (lane < chunks) {
                    const uint2 d = *reinterpret_cast<const uint2 *>(P.src + (int64_t)y * P.src_stride + xb);
                    if constexpr (GFW_LUT_TILE == 3) {
                        E *dst = reinterpret_cast<E *>(tile) + r * GFW_TILE_W + lane * PXC;
                        T px[PXC];
                        __builtin_memcpy(px, &d, 8);
                        #pragma unroll
                        for (int i = 0; i < PXC; ++i) dst[i] = (float)px[i];
                    } else {
                        tile[r * CH + lane] = d;
                    }
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        }
    }
    #pragma unroll
    for (int q = 0; q < NS; ++q) {
        if (!need[q]) continue;
        const bool fit = have && inside[q] && (b[q].sx - x0 + I <= valid_w) && (b[q].sy - y0 + I <= GFW_TILE_H);
        if (fit) {
            const E *t0 = reinterpret_cast<const E *>(tile) + (b[q].sy - y0) * GFW_TILE_W + (b[q].sx - x0);
            float cx[I];
            #pragma unroll
            for (int i = 0; i < I; ++i) cx[i] = b[q].tx[i];
            float s1 = 0.0f;
            #pragma unroll
            for (int yp = 0; yp < I; ++yp) {
                float xs = 0.0f;
                #pragma unroll
                for (int xp = 0; xp < I; ++xp) xs = xs + (float)t0[yp * GFW_TILE_W + xp] * cx[xp];
                s1 = s1 + xs * b[q].ty[yp];
            }
            const float o = fminf(s1, limit);
            store_px<T, 1>(P.dst, oy[q] * P.dst_stride + ox[q] * (int)sizeof(T), &o);
        } else {
            sample_store<T, 1, I>(u[q], v[q], ok[q], P, bg, limit, ox[q], oy[q], lut);
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");              // the next call overwrites the tile
}
// The same for bilinear taps (the north-star's "source tile in LDS" for the hot configuration): 8 rows of the wave's bounding box.
// Staged to be measured, not expected to win — the bilinear kernel is bound by instruction issue, not by its eight gathers per
// lane-row, and the copy adds ~25 instructions per plane and row.
constexpr int GFW_TILE2_H = 8;
template <typename T, int NS>
__device__ __forceinline__ void tile_sample_store2(const GfwYuvPlane &P, const float *u, const float *v, const bool *ok, const bool *need,
                                                   const int *ox, const int *oy, const float *bg, float limit,
                                                   uint2 *tile, int *org, int lane) {
    constexpr int TWB = GFW_TILE_W * (int)sizeof(T), CH = TWB / 8;
    Bins2 b[NS];
    bool inside[NS];
    #pragma unroll
    for (int q = 0; q < NS; ++q) {
        inside[q] = false;
        b[q] = Bins2{0, 0, 0.0f, 0.0f, 0.0f, 0.0f, 0u, 0u};
        if (need[q] && ok[q]) { b[q] = make_bins2(u[q], v[q]); inside[q] = (unsigned)b[q].sx < (unsigned)(P.w - 1) && (unsigned)b[q].sy < (unsigned)(P.h - 1); }
    }
    const bool all_lanes = __builtin_amdgcn_read_exec() == ~0ull;
    const bool aligned = ((P.src_stride & 3) == 0) && (((uintptr_t)P.src & 3u) == 0);
    bool have = false;
    int x0 = 0, y0 = 0, valid_w = 0;
    if (all_lanes && aligned) {
        if (lane == 0) { org[0] = 0x7fffffff; org[1] = 0x7fffffff; }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        #pragma unroll
        for (int q = 0; q < NS; ++q) if (inside[q]) { atomicMin(&org[0], b[q].sx); atomicMin(&org[1], b[q].sy); }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        const int mx = __builtin_amdgcn_readfirstlane(org[0]);
        y0 = __builtin_amdgcn_readfirstlane(org[1]);
        if (mx != 0x7fffffff) {
            have = true;
            x0 = mx & ~(int)(4 / (int)sizeof(T) - 1);
            const int row_bytes = P.w * (int)sizeof(T), x0b = x0 * (int)sizeof(T);
            const int chunks = min(CH, (row_bytes - x0b) / 8);
            valid_w = chunks * (8 / (int)sizeof(T));
            const int xb = x0b + 8 * lane;
            #pragma unroll 1
            for (int r = 0; r < GFW_TILE2_H; ++r) {
                const int y = y0 + r;
                if (y >= P.h) break;
                if (lane < chunks) tile[r * CH + lane] = *reinterpret_cast<const uint2 *>(P.src + (int64_t)y * P.src_stride + xb);
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        }
    }
    #pragma unroll
    for (int q = 0; q < NS; ++q) {
        if (!need[q]) continue;
        const bool fit = have && inside[q] && (b[q].sx - x0 + 2 <= valid_w) && (b[q].sy - y0 + 2 <= GFW_TILE2_H);
        if (fit) {
            const T *t0 = reinterpret_cast<const T *>(tile) + (b[q].sy - y0) * GFW_TILE_W + (b[q].sx - x0);
            // taps_inside2, integer pixels: the leading zero-adds of the reference are exact identities (every tap >= +0)
            const float xs0 = (float)t0[0] * b[q].cx0 + (float)t0[1] * b[q].cx1;
            const float xs1 = (float)t0[GFW_TILE_W] * b[q].cx0 + (float)t0[GFW_TILE_W + 1] * b[q].cx1;
            const float o = fminf(xs0 * b[q].cy0 + xs1 * b[q].cy1, limit);
            store_px<T, 1>(P.dst, oy[q] * P.dst_stride + ox[q] * (int)sizeof(T), &o);
        } else {
            sample_store2<T, 1>(u[q], v[q], ok[q], P, bg, limit, ox[q], oy[q], nullptr);
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
}
#endif   // GFW_LUT_TILE

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

// Byte-offset addressing from a uniform base: a 32-bit lane offset on top of a scalar base register pair.
template <typename V>
__device__ __forceinline__ V hot_ld(const void *base, uint32_t byte_off) { return *reinterpret_cast<const V *>(reinterpret_cast<const uint8_t *>(base) + byte_off); }
// First pass of the lane's two horizontally adjacent pixels at once (packed): the certified table-driven row pick of
// pass1_fast, element-wise.  good[i] false -> the exact path decides that pixel's row.
__device__ __forceinline__ void hot_pass1_pair(gfw_f2 ox, float oy, const Mid &M, const P1 &Q, const float2 *tab, bool hrs, float rl2,
                                               int &sy0, int &sy1, bool &good0, bool &good1, gfw_f2 &v_out, unsigned long long *aud) {
    const gfw_f2 oyv = {oy, oy};
    const gfw_f2 X = __builtin_elementwise_fma(oyv, gfw_f2{M.m1, M.m1}, __builtin_elementwise_fma(ox, gfw_f2{M.m0, M.m0}, gfw_f2{M.m2, M.m2}));
    const gfw_f2 Y = __builtin_elementwise_fma(oyv, gfw_f2{M.m4, M.m4}, __builtin_elementwise_fma(ox, gfw_f2{M.m3, M.m3}, gfw_f2{M.m5, M.m5}));
    const gfw_f2 W = __builtin_elementwise_fma(oyv, gfw_f2{M.m7, M.m7}, __builtin_elementwise_fma(ox, gfw_f2{M.m6, M.m6}, gfw_f2{M.m8, M.m8}));
    const gfw_f2 rw = {gfw_hw_rcp(W.x), gfw_hw_rcp(W.y)};
    const gfw_f2 a = X * rw, b = Y * rw;
    const gfw_f2 rho = __builtin_elementwise_fma(a, a, b * b);
    bool g0 = (W.x > 0.0009765625f) & (rho.x < Q.rho_max), g1 = (W.y > 0.0009765625f) & (rho.y < Q.rho_max);
    if (rl2 > 0.0f) {                                              // :139 — decide only when clear of the boundary
        const gfw_f2 lhs = __builtin_elementwise_fma(X, X, Y * Y), rhs = W * (rl2 * 0.9999f);
        g0 &= lhs.x < rhs.x; g1 &= lhs.y < rhs.y;
    }
    const gfw_f2 tpos = gfw_f2{fminf(fmaxf(rho.x, 0.0f), Q.rho_max), fminf(fmaxf(rho.y, 0.0f), Q.rho_max)} * Q.rho_scale;
    const gfw_f2 ti = {floorf(tpos.x), floorf(tpos.y)};
    if (aud && !((int)ti.x >= 0 && (int)ti.x <= GFW_P1_TABLE_N && (int)ti.y >= 0 && (int)ti.y <= GFW_P1_TABLE_N)) atomicAdd(&aud[5], 1ull);
    const float2 e0 = hot_ld<float2>(tab, (uint32_t)(int)ti.x * 8u), e1 = hot_ld<float2>(tab, (uint32_t)(int)ti.y * 8u);
    const gfw_f2 s = __builtin_elementwise_fma(tpos - ti, gfw_f2{e0.y, e1.y}, gfw_f2{e0.x, e1.x});
    const gfw_f2 v = __builtin_elementwise_fma((hrs ? a : b) * s, gfw_f2{Q.f, Q.f}, gfw_f2{Q.c, Q.c});
    v_out = v;
    const gfw_f2 g = v - 0.5f;
    const gfw_f2 d = g - gfw_f2{rintf(g.x), rintf(g.y)};           // distance of v to the nearest half-integer
    const bool out0 = !(v.x > -0.25f) | !(v.x < Q.lim + 0.25f), out1 = !(v.y > -0.25f) | !(v.y < Q.lim + 0.25f);   // there the clamp decides
    good0 = g0 & (out0 | (fabsf(d.x) > Q.eps)) & (v.x == v.x);
    good1 = g1 & (out1 | (fabsf(d.y) > Q.eps)) & (v.y == v.y);
    sy0 = max(min(gfw_f2i(rintf(v.x)), (int)Q.lim), 0);
    sy1 = max(min(gfw_f2i(rintf(v.y)), (int)Q.lim), 0);
}

#if GFW_TIMELINE
__device__ unsigned long long gfw_tl[8192 * 8];
#endif
template <int MODEL, typename T, int N0, int I, int DW, int DH, bool INTERLEAVED_UV, int RB, bool FAST1, bool AUDIT>
// Register budget: the specialised-fisheye instantiations are held to GFW_WAVES_PER_EU waves per SIMD.  The generic-model ones
// (every other lens, digital lenses, refraction, IBIS/OIS, lens-correction blend) would pay for that budget with 450-840 bytes
// of scratch per lane, and left alone they take up to 277 VGPRs (one wave per SIMD); three waves per SIMD (168 VGPRs) holds them
// with 0-250 bytes of scratch (tools/kernel_resources.py).  Not yet measured against 4 (128 VGPRs, 110-500 bytes).
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(MODEL == GFW_MODEL_OPENCV_FISHEYE ? GFW_WAVES_PER_EU : GFW_GENERIC_WAVES_PER_EU, 8))) void gfw_yuv_kernel(const GfwYuvArgs A) {
    // tile = 64 x 4 lanes; each lane owns RB vertically stacked DW x DH luma blocks (+ their chroma sites).
    constexpr int NPX = DW * DH;
    constexpr int QCAP = 128 * NPX;                  // a wave adds at most 64*NPX entries per row; flushed at half full
    static_assert(RB * NPX <= 64, "slot index must fit the 6 low bits of q_dst");
    __shared__ float q_x[FAST1 ? 4 : 1][FAST1 ? QCAP : 1], q_y[FAST1 ? 4 : 1][FAST1 ? QCAP : 1];
    __shared__ unsigned short q_dst[FAST1 ? 4 : 1][FAST1 ? QCAP : 1];           // (owner lane << 6) | slot in s_rows
    __shared__ unsigned q_n[4];
    __shared__ int s_rows[RB * NPX][256];                                        // phase-1 rows, one column per lane
#if GFW_LDS_MATRICES
    constexpr bool MWIN = FAST1 && !AUDIT;
    constexpr int MW_ROWS = 32, MW_PITCH = 12;                                   // 32 rows of 12 floats (m0..m8 + 3 of padding: 16-byte rows)
    __shared__ __attribute__((aligned(16))) float s_mat[MWIN ? 4 : 1][MWIN ? MW_ROWS * MW_PITCH : 4];
    __shared__ int s_mrow[4];
#endif
    __shared__ float s_lut[I == 2 ? 1 : 448];                                    // bicubic / Lanczos4 tap table
#if GFW_LUT_TILE
    // GFW_LUT_TILE = 1: bicubic / Lanczos4 only; 2: bilinear as well (the A/B the north-star asks for)
    constexpr bool TILE = MODEL == GFW_MODEL_OPENCV_FISHEYE && (I != 2 || GFW_LUT_TILE == 2) && !AUDIT && !is_f32<T>::value && N0 == 1 && !INTERLEAVED_UV && NPX <= 2;
    __shared__ uint2 s_tile[TILE ? 4 : 1][TILE ? ((I == 2 ? GFW_TILE2_H : GFW_TILE_H) * GFW_TILE_W * (int)(GFW_LUT_TILE == 3 && I != 2 ? sizeof(float) : sizeof(T))) / 8 : 1];
    __shared__ int s_org[4][2];
#endif
    const int wave = threadIdx.y, lane = threadIdx.x, tid = wave * 64 + lane;
#if GFW_ATAN_TABLE
    if (MODEL == GFW_MODEL_OPENCV_FISHEYE) gfw_atan_lds_init(tid);
#endif
    if (I != 2) {
        for (int i = tid; i < 448; i += 256) s_lut[i] = GFW_COEFFS[i];
        __syncthreads();
    }
    const bool two_pass = A.matrix_count > 1 && !(A.ablate & 1);
    const bool hrs = A.hrs != 0;

    // uniform floats of the pixel loops, pinned in VGPRs once per wave
    Lens L;
    L.f0 = vu_lens(vu(A.f[0])); L.f1 = vu_lens(vu(A.f[1])); L.c0 = vu_lens(vu(A.c[0])); L.c1 = vu_lens(vu(A.c[1]));
    L.k0 = vu_lens(vu(A.k[0])); L.k1 = vu_lens(vu(A.k[1])); L.k2 = vu_lens(vu(A.k[2])); L.k3 = vu_lens(vu(A.k[3]));
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
#if GFW_XCD_CHUNK > 0
    const int per_xcd = ((((n_tiles + GFW_XCD_CHUNK - 1) / GFW_XCD_CHUNK) + 7) >> 3) * GFW_XCD_CHUNK;
#define GFW_XCD_TILE(l) ((((l) / GFW_XCD_CHUNK) * 8 + xcd) * GFW_XCD_CHUNK + (l) % GFW_XCD_CHUNK)
#else
    const int per_xcd = (n_tiles + 7) >> 3;
#define GFW_XCD_TILE(l) (xcd * per_xcd + (l))
#endif
    const int wg_per_xcd = (int)gridDim.x >> 3;
    const int xcd = (int)blockIdx.x & 7;
#if GFW_TIMELINE
    const unsigned long long tl_start = wall_clock64();
    unsigned long long tl_p1 = 0, tl_p3 = 0, tl_units = 0;
#endif
#if GFW_PRIO_MODE
    auto set_prio = [](int p) {                       // s_setprio takes an immediate
        if (p <= 0) __builtin_amdgcn_s_setprio(0); else if (p == 1) __builtin_amdgcn_s_setprio(1);
        else if (p == 2) __builtin_amdgcn_s_setprio(2); else __builtin_amdgcn_s_setprio(3);
    };
    const int prio_age = (((int)blockIdx.x >> 3) * GFW_PRIO_AGE_ROWS) / (wg_per_xcd > 0 ? wg_per_xcd : 1);
    int tiles_left = 0;
    for (int tb = (int)blockIdx.x >> 3; tb < per_xcd && GFW_XCD_TILE(tb) < n_tiles; tb += wg_per_xcd) ++tiles_left;
#endif
    for (int tb = (int)blockIdx.x >> 3; tb < per_xcd; tb += wg_per_xcd) {
        const int t = GFW_XCD_TILE(tb);
        if (t >= n_tiles) break;
#if GFW_PRIO_MODE == 2
        set_prio(tiles_left < 3 ? tiles_left : 3);
#elif GFW_PRIO_MODE == 1
        set_prio((tiles_left * RB + prio_age) / GFW_PRIO_DIV);
#endif
        const int ty = t / A.tiles_x, tx = t - ty * A.tiles_x;
        const int cx = tx * 64 + lane;
        const int cy0 = (ty * 4 + wave) * RB;            // first chroma-site row of this lane
        const bool lane_ok = cx < A.cw;

#if GFW_TIMELINE
        const unsigned long long tl_a = __builtin_readcyclecounter();
#endif
        // ---- phase 1: rolling-shutter row of every luma pixel of this lane ----------------------------
        if (two_pass) {
            if (FAST1) {
                if (lane == 0) q_n[wave] = 0;
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            }
            #pragma unroll 1
            for (int r = 0; r < RB; ++r) {
              if (GFW_PASS1_PAIR && FAST1 && !AUDIT && NPX == 2 && DW == 2) {
                // the lane's two horizontally adjacent pixels at once, packed (hot_pass1_pair): same certificate, element-wise
                const int lx0 = cx * 2, ly = cy0 + r;
                const gfw_f2 oxp = {(float)lx0 + L.t2x, (float)(lx0 + 1) + L.t2x};
                const float oy = (float)ly + L.t2y;
                int sy0, sy1; bool g0, g1; gfw_f2 v_fast;
                hot_pass1_pair(oxp, oy, M, Q, A.p1_table, hrs, L.rl2, sy0, sy1, g0, g1, v_fast, nullptr);
                const bool in0 = lane_ok && lx0 < A.out_w && ly < A.out_h, in1 = lane_ok && lx0 + 1 < A.out_w && ly < A.out_h;
                if (in0 && !g0) {
                    const unsigned slot = atomicAdd(&q_n[wave], 1u);
                    q_x[wave][slot] = oxp.x; q_y[wave][slot] = oy; q_dst[wave][slot] = (unsigned short)((lane << 6) | (r * NPX));
                }
                if (in1 && !g1) {
                    const unsigned slot = atomicAdd(&q_n[wave], 1u);
                    q_x[wave][slot] = oxp.y; q_y[wave][slot] = oy; q_dst[wave][slot] = (unsigned short)((lane << 6) | (r * NPX + 1));
                }
                s_rows[r * NPX][tid] = in0 ? sy0 : 0;
                s_rows[r * NPX + 1][tid] = in1 ? sy1 : 0;
              } else {
                #pragma unroll (NPX <= 2 ? NPX : 1)
                for (int k = 0; k < NPX; ++k) {
                    const int i = k % DW, j = k / DW;
                    const int lx = cx * DW + i, ly = (cy0 + r) * DH + j;
                    int sy = 0;
                    if (lane_ok && lx < A.out_w && ly < A.out_h) {
                        float ox = (float)lx + L.t2x, oy = (float)ly + L.t2y;
                        if (MODEL != GFW_MODEL_OPENCV_FISHEYE && (A.extras & 8)) gfw_lens_correction_blend<MODEL>(ox, oy, A.kp, A.common);   // :429-460
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

#if GFW_LDS_MATRICES
        int mw0 = 0; bool mw_on = false;
        if constexpr (MWIN) if (two_pass) {
            // smallest row index any pixel of this wave's tile uses; the window holds the 32 rows from there
            int my_min = 0x7fffffff;
            #pragma unroll 1
            for (int q = 0; q < RB * NPX; ++q) {
                const int r = q / NPX, k = q - r * NPX;
                const int lx = cx * DW + k % DW, ly = (cy0 + r) * DH + k / DW;
                if (lane_ok && lx < A.out_w && ly < A.out_h) my_min = min(my_min, s_rows[q][tid]);
            }
            if (lane == 0) s_mrow[wave] = 0x7fffffff;
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            if (my_min != 0x7fffffff) atomicMin(&s_mrow[wave], my_min);
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            const int m = __builtin_amdgcn_readfirstlane(s_mrow[wave]);
            if (m != 0x7fffffff) {
                mw0 = min(m, A.matrix_count - 1);
                #pragma unroll
                for (int e = lane; e < MW_ROWS * MW_PITCH; e += 64) {
                    const int row = e / MW_PITCH, col = e - row * MW_PITCH;
                    s_mat[wave][e] = A.matrices[(size_t)min(mw0 + row, A.matrix_count - 1) * GFW_MAT_STRIDE + col];
                }
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                mw_on = true;
            }
        }
#endif
#if GFW_TIMELINE
        const unsigned long long tl_b = __builtin_readcyclecounter();
#endif
        // ---- phase 3: exact projection with the row's own matrix, then taps ---------------------------
        if (lane_ok) {
            #pragma unroll 1
            for (int r = 0; r < RB; ++r) {
#if GFW_PRIO_MODE == 1
                set_prio(((tiles_left * RB) - r + prio_age) / GFW_PRIO_DIV);
#endif
                const int cy = cy0 + r;
                if (cy >= A.ch) break;
                float u0 = 0.0f, v0 = 0.0f; bool ok0 = false;
#if GFW_LUT_TILE
                float tl_u[NPX], tl_v[NPX]; bool tl_ok[NPX], tl_need[NPX]; int tl_x[NPX], tl_y[NPX];
                #pragma unroll
                for (int k = 0; k < NPX; ++k) { tl_u[k] = 0.0f; tl_v[k] = 0.0f; tl_ok[k] = false; tl_need[k] = false; tl_x[k] = 0; tl_y[k] = 0; }
#endif
                #pragma unroll (NPX <= 2 ? NPX : 1)
                for (int k = 0; k < NPX; ++k) {
                    const int i = k % DW, j = k / DW;
                    const int lx = cx * DW + i, ly = cy * DH + j;
                    if (lx >= A.out_w || ly >= A.out_h) continue;
                    float ox = (float)lx + L.t2x, oy = (float)ly + L.t2y;
                    if (MODEL != GFW_MODEL_OPENCV_FISHEYE && (A.extras & 8)) gfw_lens_correction_blend<MODEL>(ox, oy, A.kp, A.common);       // :429-460
                    const int sy = two_pass ? s_rows[r * NPX + k][tid] : default_row<MODEL>(ox, oy, A);
                    GfwPt p;
                    if (A.ablate & 8) { p.x = ox * 0.5f; p.y = oy * 0.5f; p.ok = true; }              // timing ablation only
                    else {
                        const int row = min(sy, A.matrix_count - 1);
                        if (AUDIT && (unsigned)row >= (unsigned)A.matrix_count) atomicAdd(&A.audit[5], 1ull);
#if GFW_LDS_MATRICES
                        if constexpr (MWIN) {                                  // one projection, its nine matrix entries from the window or from memory
                            const float *g = A.matrices + (size_t)row * GFW_MAT_STRIDE;
                            float4 ma, mb; float m8;
                            if (mw_on && (unsigned)(row - mw0) < (unsigned)MW_ROWS) {
                                const float *w = &s_mat[wave][(row - mw0) * MW_PITCH];
                                ma = *reinterpret_cast<const float4 *>(w); mb = *reinterpret_cast<const float4 *>(w + 4); m8 = w[8];
                            } else {
                                ma = *reinterpret_cast<const float4 *>(g); mb = *reinterpret_cast<const float4 *>(g + 4); m8 = g[8];
                            }
                            p = rd<MODEL>(ox, oy, ma, mb, m8, g + 8, L, A);
                        } else
#endif
                        p = rd_row<MODEL>(ox, oy, row, L, A);
                    }
                    if ((A.background_mode == 1 || A.background_mode == 2) && p.ok) {                      // cpu_undistort.rs:495-509 (edge repeat / edge mirror)
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
#if GFW_STAGED_FUSED
                    if (MODEL != GFW_MODEL_OPENCV_FISHEYE && (A.extras & 16) && p.ok) {          // background mode 3: two samples, blended (:576-613)
                        feather_store<T, N0, I, true>(p.x, p.y, feather_of(p.x, p.y, A), A.pl[0], bg_y, lim_y, MP.mul_lx, MP.mul_ly, MP, lx, ly, s_lut);
                        continue;
                    }
#endif
                    const float lu = map_c<MODEL != GFW_MODEL_OPENCV_FISHEYE>(p.x, MP.mul_lx, MP.den_x, MP.rcp_x), lv = map_c<MODEL != GFW_MODEL_OPENCV_FISHEYE>(p.y, MP.mul_ly, MP.den_y, MP.rcp_y);   // cpu_undistort.rs:511-514
                    if (A.ablate & 2) { if (lane == 99) A.pl[0].dst[0] = (uint8_t)(lu + lv); continue; }  // timing ablation only
#if GFW_LUT_TILE
                    if constexpr (TILE) { tl_u[k] = lu; tl_v[k] = lv; tl_ok[k] = p.ok; tl_need[k] = true; tl_x[k] = lx; tl_y[k] = ly; }
                    else
#endif
                    if (I == 2) sample_store2<T, N0>(lu, lv, p.ok, A.pl[0], bg_y, lim_y, lx, ly, AUDIT ? A.audit : nullptr);
                    else sample_store<T, N0, I>(lu, lv, p.ok, A.pl[0], bg_y, lim_y, lx, ly, s_lut);
                }
#if GFW_LUT_TILE
                if constexpr (TILE) {
                    if (!(A.ablate & 2)) {
                        if constexpr (I == 2) tile_sample_store2<T, NPX>(A.pl[0], tl_u, tl_v, tl_ok, tl_need, tl_x, tl_y, bg_y, lim_y, s_tile[wave], s_org[wave], lane);
                        else tile_sample_store<T, I, NPX>(A.pl[0], tl_u, tl_v, tl_ok, tl_need, tl_x, tl_y, bg_y, lim_y, s_tile[wave], s_org[wave], s_lut, lane);
                    }
                    if (A.nplanes == 3 && !(A.ablate & 4)) {               // planar U and V: same bins, one tile each
                        const float cu = map_c<false>(u0, MP.mul_cx, MP.den_x, MP.rcp_x), cv = map_c<false>(v0, MP.mul_cy, MP.den_y, MP.rcp_y);
                        const bool need1 = true;
                        if constexpr (I == 2) {
                            tile_sample_store2<T, 1>(A.pl[1], &cu, &cv, &ok0, &need1, &cx, &cy, bg_c, lim_u, s_tile[wave], s_org[wave], lane);
                            tile_sample_store2<T, 1>(A.pl[2], &cu, &cv, &ok0, &need1, &cx, &cy, &bg_v, lim_v, s_tile[wave], s_org[wave], lane);
                        } else {
                            tile_sample_store<T, I, 1>(A.pl[1], &cu, &cv, &ok0, &need1, &cx, &cy, bg_c, lim_u, s_tile[wave], s_org[wave], s_lut, lane);
                            tile_sample_store<T, I, 1>(A.pl[2], &cu, &cv, &ok0, &need1, &cx, &cy, &bg_v, lim_v, s_tile[wave], s_org[wave], s_lut, lane);
                        }
                        continue;
                    }
                }
#endif
#if GFW_STAGED_FUSED
                if (MODEL != GFW_MODEL_OPENCV_FISHEYE && (A.extras & 16) && ok0 && A.nplanes > 1) {  // background mode 3 for the chroma site
                    const Feather f = feather_of(u0, v0, A);
                    if (INTERLEAVED_UV) feather_store<T, 2, I, true>(u0, v0, f, A.pl[1], bg_c, lim_u, MP.mul_cx, MP.mul_cy, MP, cx, cy, s_lut);
                    else for (int pi = 1; pi < A.nplanes; ++pi)
                        feather_store<T, 1, I, true>(u0, v0, f, A.pl[pi], A.pl[pi].bg, A.pl[pi].limit, MP.mul_cx, MP.mul_cy, MP, cx, cy, s_lut);
                } else
#endif
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
#if GFW_PRIO_MODE
        --tiles_left;
#endif
#if GFW_TIMELINE
        { const unsigned long long tl_c = __builtin_readcyclecounter(); tl_p1 += tl_b - tl_a; tl_p3 += tl_c - tl_b; tl_units += (unsigned long long)RB; }
#endif
    }
#if GFW_TIMELINE
    if (lane == 0) {       // per wave: start, end (100 MHz device clock), phase clocks, lane-rows, HW_ID, XCC_ID, workgroup
        unsigned long long *o = gfw_tl + ((size_t)blockIdx.x * 4 + wave) * 8;
        o[0] = tl_start; o[1] = wall_clock64(); o[2] = tl_p1; o[3] = tl_p3; o[4] = tl_units;
        o[5] = __builtin_amdgcn_s_getreg(4 | (31 << 11)); o[6] = __builtin_amdgcn_s_getreg(20 | (31 << 11)); o[7] = blockIdx.x;
    }
#endif
#undef GFW_XCD_TILE
}


// =====================================================================================================================
// gfw_hot_kernel — the production configuration (opencv_fisheye, bilinear, 8/16-bit planar or semi-planar 4:2:2 / 4:2:0,
// solid background) with a CERTIFIED SECOND PASS and integer-dot-product taps.
//
// Second pass.  The head of rotate_and_distort is evaluated exactly as the reference does (X, Y, W and the two correctly
// rounded divisions: cpu_undistort.rs:134-137, opencv_fisheye.rs:73), so a = X/W and b = Y/W are the reference's bits.  The
// expensive middle — r = sqrt(a^2+b^2), theta = atanf(r), the theta_d polynomial and the division theta_d/r
// (opencv_fisheye.rs:77-93), two thirds of the exact projection's instructions — is replaced by s~ = S(rho~), rho~ = a^2+b^2,
// read from the first pass's table, together with a RELATIVE bound kappa on |s~ - s_ref| (derivation in DESIGN.md section 2b:
// every rounding of the reference's chain and of this one is accounted for; atanf's error constant is measured over all
// positive floats, tests/test_math_host.py).  The rest of the chain — a*s, *f, +c, the source_rect map, *32, round — is made of
// monotone non-decreasing steps (round-to-nearest is monotone, f > 0, mul > 0), so the 1/32-pixel bin is a monotone step
// function of s: it is evaluated at s_lo = s~(1-kappa) and s_hi = s~(1+kappa) with the reference's own operations (packed,
// two values per instruction), and if both ends land in the same bin — for x and y, and for the chroma site's bins when the
// pixel carries one — that bin IS the reference's.  No error analysis of the tail is involved.  Pixels whose ends disagree
// (a few percent) are queued per wave and resolved densely by the exact projection, like the first pass's rejects.
// The audit instantiation recomputes the exact bins of EVERY accepted pixel and counts disagreements (there must be none).
//
// Taps.  A bilinear sample of an integer plane is sum = RN(RN(xs0*cy0) + RN(xs1*cy1)) with xs = p0*(1-k/32) + p1*k/32 exact
// (cpu_undistort.rs:392-411).  Here xs*32 = p0*(32-k) + p1*k is ONE integer dot instruction on the raw loaded word
// (v_dot2_u32_u16 / v_dot4_u32_u8), converted exactly (< 2^22), multiplied by the integer y weight and scaled by 2^-10 at the
// end — power-of-two scaling commutes with round-to-nearest, so the two roundings are the reference's.
// =====================================================================================================================
#if GFW_FRAME_TAPS == 2 && GFW_FRAME_KIND != 4 && !GFW_HOT_ONLY
__device__ __forceinline__ Bins2 hot_bins2(int bx, int by) {
    Bins2 b;
    b.sx = bx >> 5; b.sy = by >> 5;
    b.kx = (uint32_t)bx & 31u; b.ky = (uint32_t)by & 31u;
    b.cx1 = (float)(bx & 31) * 0.03125f; b.cx0 = 1.0f - b.cx1;
    b.cy1 = (float)(by & 31) * 0.03125f; b.cy0 = 1.0f - b.cy1;
    return b;
}
// One single-channel sample from its 1/32-pixel bins (bx, by) = (round(u*32), round(v*32)): the pixel value as an integer.
template <typename T>
__device__ __forceinline__ uint32_t hot_sample(const GfwYuvPlane &P, int bx, int by, float bg, float limit, unsigned long long *aud) {
    typedef HotTap<T, false> Tap;
    const int sx = bx >> 5, sy = by >> 5;
    if (__builtin_expect((unsigned)sx < (unsigned)(P.w - 1) && (unsigned)sy < (unsigned)(P.h - 1), 1)) {
        const uint32_t off = (uint32_t)sy * (uint32_t)P.src_stride + (uint32_t)sx * (uint32_t)sizeof(T);
        if (!range_ok(aud, off, Tap::BYTES, P.src_len) || !range_ok(aud, (int64_t)off + P.src_stride, Tap::BYTES, P.src_len)) return 0u;
        const uint32_t r0 = Tap::load(P.src, off), r1 = Tap::load(P.src, off + (uint32_t)P.src_stride);
        const uint32_t w = Tap::wpack((uint32_t)bx & 31u);
        return hot_blend(Tap::dot(r0, w), Tap::dot(r1, w), (uint32_t)by & 31u, limit);
    }
    float o;
    taps_edge2<T, 1>(P.src, P.src_stride, hot_bins2(bx, by), P.w, P.h, &bg, limit, &o);
    return gfw_f2u_sat(o, sizeof(T) == 1 ? 255.0f : 65535.0f);
}
// The chroma site: two planar planes of identical geometry (U, V) sharing bins, or one interleaved UV plane.
template <typename T, bool INTERLEAVED_UV>
__device__ __forceinline__ void hot_sample_uv(const GfwYuvPlane &PU, const GfwYuvPlane &PV, int bx, int by, float bg_u, float bg_v, float lim_u, float lim_v,
                                              uint32_t &ou, uint32_t &ov, unsigned long long *aud) {
    const int sx = bx >> 5, sy = by >> 5;
    const bool inside = (unsigned)sx < (unsigned)(PU.w - 1) && (unsigned)sy < (unsigned)(PU.h - 1);
    ou = 0u; ov = 0u;
    if (INTERLEAVED_UV) {
        typedef HotTap<T, true> Tap;
        if (__builtin_expect(inside, 1)) {
            const uint32_t off = (uint32_t)sy * (uint32_t)PU.src_stride + (uint32_t)sx * (uint32_t)(2 * sizeof(T));
            if (!range_ok(aud, off, Tap::BYTES, PU.src_len) || !range_ok(aud, (int64_t)off + PU.src_stride, Tap::BYTES, PU.src_len)) return;
            const auto r0 = Tap::load(PU.src, off), r1 = Tap::load(PU.src, off + (uint32_t)PU.src_stride);
            const uint32_t w = Tap::wpack((uint32_t)bx & 31u);
            uint32_t u0, v0, u1, v1;
            Tap::dot(r0, w, u0, v0); Tap::dot(r1, w, u1, v1);
            ou = hot_blend(u0, u1, (uint32_t)by & 31u, lim_u);
            ov = hot_blend(v0, v1, (uint32_t)by & 31u, lim_u);
        } else {
            float bg[2] = {bg_u, bg_v}, o[2];
            taps_edge2<T, 2>(PU.src, PU.src_stride, hot_bins2(bx, by), PU.w, PU.h, bg, lim_u, o);
            ou = gfw_f2u_sat(o[0], sizeof(T) == 1 ? 255.0f : 65535.0f); ov = gfw_f2u_sat(o[1], sizeof(T) == 1 ? 255.0f : 65535.0f);
        }
    } else {
        typedef HotTap<T, false> Tap;
        if (__builtin_expect(inside, 1)) {
            const uint32_t off = (uint32_t)sy * (uint32_t)PU.src_stride + (uint32_t)sx * (uint32_t)sizeof(T);
            const int top = PU.src_len < PV.src_len ? PU.src_len : PV.src_len;
            if (!range_ok(aud, off, Tap::BYTES, top) || !range_ok(aud, (int64_t)off + PU.src_stride, Tap::BYTES, top)) return;
            const uint32_t a0 = Tap::load(PU.src, off), a1 = Tap::load(PU.src, off + (uint32_t)PU.src_stride);
            const uint32_t b0 = Tap::load(PV.src, off), b1 = Tap::load(PV.src, off + (uint32_t)PU.src_stride);
            const uint32_t w = Tap::wpack((uint32_t)bx & 31u);
            ou = hot_blend(Tap::dot(a0, w), Tap::dot(a1, w), (uint32_t)by & 31u, lim_u);
            ov = hot_blend(Tap::dot(b0, w), Tap::dot(b1, w), (uint32_t)by & 31u, lim_v);
        } else {
            float o;
            const Bins2 b = hot_bins2(bx, by);
            taps_edge2<T, 1>(PU.src, PU.src_stride, b, PU.w, PU.h, &bg_u, lim_u, &o); ou = gfw_f2u_sat(o, sizeof(T) == 1 ? 255.0f : 65535.0f);
            taps_edge2<T, 1>(PV.src, PU.src_stride, b, PU.w, PU.h, &bg_v, lim_v, &o); ov = gfw_f2u_sat(o, sizeof(T) == 1 ? 255.0f : 65535.0f);
        }
    }
}
template <typename T, bool INTERLEAVED_UV>
__device__ __forceinline__ void hot_store_uv(const GfwYuvPlane &PU, const GfwYuvPlane &PV, int cx, int cy, uint32_t ou, uint32_t ov, unsigned long long *aud) {
    if (INTERLEAVED_UV) {
        const uint32_t doff = (uint32_t)cy * (uint32_t)PU.dst_stride + (uint32_t)cx * (uint32_t)(2 * sizeof(T));
        if (!range_ok(aud, doff, 2 * sizeof(T), PU.dst_len)) return;
        if (sizeof(T) == 2) *reinterpret_cast<uint32_t *>(PU.dst + doff) = ou | (ov << 16);
        else *reinterpret_cast<uint16_t *>(PU.dst + doff) = (uint16_t)(ou | (ov << 8));
    } else {
        const uint32_t doff = (uint32_t)cy * (uint32_t)PU.dst_stride + (uint32_t)cx * (uint32_t)sizeof(T);
        if (!range_ok(aud, doff, sizeof(T), PU.dst_len < PV.dst_len ? PU.dst_len : PV.dst_len)) return;
        *reinterpret_cast<T *>(PU.dst + doff) = (T)ou;
        *reinterpret_cast<T *>(PV.dst + doff) = (T)ov;
    }
}

struct HotQ { float rho_max, rho_scale, kappa; };
// Uniform constants of the tail, as (x, y) pairs: the packed instructions take them as natural 64-bit scalar operands.
struct HotC { gfw_f2 f, c, mul_l, mul_c, nden, rcp; };
__device__ __forceinline__ gfw_f2 hot_map2(gfw_f2 x, gfw_f2 mul, gfw_f2 nden, gfw_f2 rcp) {      // map_c on (x, y)
    const gfw_f2 a = x * mul;
    const gfw_f2 q0 = a * rcp;
    const gfw_f2 r0 = __builtin_elementwise_fma(nden, q0, a);
    return __builtin_elementwise_fma(r0, rcp, q0);
}
__device__ __forceinline__ void hot_bins(gfw_f2 xy, int &bx, int &by) {                           // round(x*32), round(y*32)
    const gfw_f2 g = xy * 32.0f;
    bx = round_i32(g.x); by = round_i32(g.y);
}

// Second pass of one pixel with the matrix row at byte offset `moff`: exact head, certified middle, exact two-point tail.
//   ok  : the reference's validity (w > 0, r_limit) — exact
//   acc : every bin below is certified; otherwise the exact path must decide
__device__ __forceinline__ void hot_project(float ox, float oy, const float *matrices, uint32_t moff, float rl2, const HotC &K, const HotQ &Q, const float2 *tab,
                                            bool with_chroma, bool &ok, bool &acc, int &bx, int &by, int &cbx, int &cby, unsigned long long *aud, int ablate = 0) {
    const float4 ma = hot_ld<float4>(matrices, moff), mb = hot_ld<float4>(matrices, moff + 16u);
    const float m8 = hot_ld<float>(matrices, moff + 32u);
    const float X = (ox * ma.x) + (oy * ma.y) + ma.z;                  // cpu_undistort.rs:134-136 (translation3d == 0)
    const float Y = (ox * ma.w) + (oy * mb.x) + mb.y;
    const float W = (ox * mb.z) + (oy * mb.w) + m8;
    ok = W > 0.0f;
    if (rl2 > 0.0f && (X * X + Y * Y) > rl2 * W) ok = false;           // :139
    const float mag = fmaxf(fmaxf(fabsf(X), fabsf(Y)), W);
    const bool lean = (mag <= 524288.0f) && (W >= 9.5367431640625e-07f);      // proven operand range of the lean divide
    gfw_f2 ab;
    { float a, b; LeanOps::div2(X, Y, W, a, b); ab.x = a; ab.y = b; }  // opencv_fisheye.rs:73 — the reference's a, b
    const float rho = __builtin_fmaf(ab.x, ab.x, ab.y * ab.y);
    const float tpos = fminf(rho, Q.rho_max) * Q.rho_scale;            // a NaN / oversized rho still indexes the table; `acc` rejects it
    const float ti = floorf(tpos);
    if (aud && !((int)ti >= 0 && (int)ti <= GFW_P1_TABLE_N)) atomicAdd(&aud[5], 1ull);
    float2 e = float2{1.0f, 0.0f};
    if (!(ablate & 16)) e = hot_ld<float2>(tab, (uint32_t)(int)ti * 8u);
    const float s = __builtin_fmaf(tpos - ti, e.y, e.x);
    const float s_lo = __builtin_fmaf(-s, Q.kappa, s), s_hi = __builtin_fmaf(s, Q.kappa, s);
    const gfw_f2 p_lo = ((ab * s_lo) * K.f) + K.c;                     // opencv_fisheye.rs:94, cpu_undistort.rs:155,167 at s_lo
    const gfw_f2 p_hi = ((ab * s_hi) * K.f) + K.c;                     // ... and at s_hi
    int bx2, by2;
    hot_bins(hot_map2(p_lo, K.mul_l, K.nden, K.rcp), bx, by);          // :511-514, :380-381
    hot_bins(hot_map2(p_hi, K.mul_l, K.nden, K.rcp), bx2, by2);
    acc = lean & (rho < Q.rho_max) & (bx == bx2) & (by == by2);
    cbx = 0; cby = 0;
    if (with_chroma) {
        hot_bins(hot_map2(p_lo, K.mul_c, K.nden, K.rcp), cbx, cby);
        hot_bins(hot_map2(p_hi, K.mul_c, K.nden, K.rcp), bx2, by2);
        acc &= (cbx == bx2) & (cby == by2);
    }
}
// ---- the lane's two horizontally adjacent pixels at once: loads of both pixels are in flight together and the head runs
// packed across the pair (same operations, same order, per element) --------------------------------------------------------
struct HotPair { bool ok[2], acc[2]; int bx[2], by[2], cbx, cby; };
__device__ __forceinline__ void hot_project_pair(gfw_f2 ox, float oy, const float *matrices, uint32_t moff0, uint32_t moff1, float rl2, const HotC &K,
                                                 const HotQ &Q, const float2 *tab, bool with_chroma, HotPair &R, unsigned long long *aud) {
    const float4 ma0 = hot_ld<float4>(matrices, moff0), mb0 = hot_ld<float4>(matrices, moff0 + 16u);
    const float4 ma1 = hot_ld<float4>(matrices, moff1), mb1 = hot_ld<float4>(matrices, moff1 + 16u);
    const float m80 = hot_ld<float>(matrices, moff0 + 32u), m81 = hot_ld<float>(matrices, moff1 + 32u);
    const gfw_f2 oyv = {oy, oy};
    const gfw_f2 X = ((ox * gfw_f2{ma0.x, ma1.x}) + (oyv * gfw_f2{ma0.y, ma1.y})) + gfw_f2{ma0.z, ma1.z};      // cpu_undistort.rs:134-136
    const gfw_f2 Y = ((ox * gfw_f2{ma0.w, ma1.w}) + (oyv * gfw_f2{mb0.x, mb1.x})) + gfw_f2{mb0.y, mb1.y};
    const gfw_f2 W = ((ox * gfw_f2{mb0.z, mb1.z}) + (oyv * gfw_f2{mb0.w, mb1.w})) + gfw_f2{m80, m81};
    R.ok[0] = W.x > 0.0f; R.ok[1] = W.y > 0.0f;
    if (rl2 > 0.0f) {                                                                                           // :139
        const gfw_f2 lhs = (X * X) + (Y * Y), rhs = W * rl2;
        R.ok[0] = R.ok[0] && !(lhs.x > rhs.x); R.ok[1] = R.ok[1] && !(lhs.y > rhs.y);
    }
    const bool lean0 = (fmaxf(fmaxf(fabsf(X.x), fabsf(Y.x)), W.x) <= 524288.0f) && (W.x >= 9.5367431640625e-07f);
    const bool lean1 = (fmaxf(fmaxf(fabsf(X.y), fabsf(Y.y)), W.y) <= 524288.0f) && (W.y >= 9.5367431640625e-07f);
    // gfw_rcp_prepare + gfw_div_prepared (gfw_fastmath.h), element-wise: the correctly rounded X/W and Y/W inside the lean range
    const gfw_f2 r0 = {gfw_hw_rcp(W.x), gfw_hw_rcp(W.y)};
    const gfw_f2 one = {1.0f, 1.0f};
    const gfw_f2 rr = __builtin_elementwise_fma(__builtin_elementwise_fma(-W, r0, one), r0, r0);
    const gfw_f2 qa = X * rr, qb = Y * rr;
    const gfw_f2 a = __builtin_elementwise_fma(__builtin_elementwise_fma(-W, qa, X), rr, qa);
    const gfw_f2 b = __builtin_elementwise_fma(__builtin_elementwise_fma(-W, qb, Y), rr, qb);
    const gfw_f2 rho = __builtin_elementwise_fma(a, a, b * b);
    const gfw_f2 tpos = gfw_f2{fminf(rho.x, Q.rho_max), fminf(rho.y, Q.rho_max)} * Q.rho_scale;   // NaN / oversized rho still index the table; acc rejects
    const gfw_f2 ti = {floorf(tpos.x), floorf(tpos.y)};
    if (aud && !((int)ti.x >= 0 && (int)ti.x <= GFW_P1_TABLE_N && (int)ti.y >= 0 && (int)ti.y <= GFW_P1_TABLE_N)) atomicAdd(&aud[5], 1ull);
    const float2 e0 = hot_ld<float2>(tab, (uint32_t)(int)ti.x * 8u), e1 = hot_ld<float2>(tab, (uint32_t)(int)ti.y * 8u);
    const gfw_f2 s = __builtin_elementwise_fma(tpos - ti, gfw_f2{e0.y, e1.y}, gfw_f2{e0.x, e1.x});
    const gfw_f2 kap = {Q.kappa, Q.kappa};
    const gfw_f2 s_lo = __builtin_elementwise_fma(-s, kap, s), s_hi = __builtin_elementwise_fma(s, kap, s);
    R.cbx = 0; R.cby = 0;
    #pragma unroll
    for (int i = 0; i < 2; ++i) {
        const gfw_f2 ab = {i ? a.y : a.x, i ? b.y : b.x};
        const float lo = i ? s_lo.y : s_lo.x, hi = i ? s_hi.y : s_hi.x;
        const gfw_f2 p_lo = ((ab * lo) * K.f) + K.c;                   // opencv_fisheye.rs:94, cpu_undistort.rs:155,167 at s_lo
        const gfw_f2 p_hi = ((ab * hi) * K.f) + K.c;                   // ... and at s_hi
        int bx2, by2;
        hot_bins(hot_map2(p_lo, K.mul_l, K.nden, K.rcp), R.bx[i], R.by[i]);          // :511-514, :380-381
        hot_bins(hot_map2(p_hi, K.mul_l, K.nden, K.rcp), bx2, by2);
        R.acc[i] = (i ? lean1 : lean0) & ((i ? rho.y : rho.x) < Q.rho_max) & (R.bx[i] == bx2) & (R.by[i] == by2);
        if (i == 0 && with_chroma) {
            hot_bins(hot_map2(p_lo, K.mul_c, K.nden, K.rcp), R.cbx, R.cby);
            hot_bins(hot_map2(p_hi, K.mul_c, K.nden, K.rcp), bx2, by2);
            R.acc[0] &= (R.cbx == bx2) & (R.cby == by2);
        }
    }
}
// Branch-free interior sample: the loads are issued whatever the lane's state (offset 0 when the taps are not all inside)
// so that every sample of the pair is in flight together; `inside` tells the caller whether the value is the real one.
template <typename T>
__device__ __forceinline__ uint32_t hot_sample_free(const GfwYuvPlane &P, int pw, int ph, int bx, int by, float limit, bool &inside, unsigned long long *aud) {
    typedef HotTap<T, false> Tap;
    const int sx = bx >> 5, sy = by >> 5;
    inside = (unsigned)sx < (unsigned)(pw - 1) && (unsigned)sy < (unsigned)(ph - 1);
    uint32_t off = inside ? (uint32_t)sy * (uint32_t)P.src_stride + (uint32_t)sx * (uint32_t)sizeof(T) : 0u;
    if (!range_ok(aud, off, Tap::BYTES, P.src_len) || !range_ok(aud, (int64_t)off + P.src_stride, Tap::BYTES, P.src_len)) off = 0u;
    const uint32_t r0 = Tap::load(P.src, off), r1 = Tap::load(P.src, off + (uint32_t)P.src_stride);
    const uint32_t w = Tap::wpack((uint32_t)bx & 31u);
    return hot_blend(Tap::dot(r0, w), Tap::dot(r1, w), (uint32_t)by & 31u, limit);
}
template <typename T, bool INTERLEAVED_UV>
__device__ __forceinline__ void hot_sample_uv_free(const GfwYuvPlane &PU, const GfwYuvPlane &PV, int bx, int by, float lim_u, float lim_v,
                                                   uint32_t &ou, uint32_t &ov, bool &inside, unsigned long long *aud) {
    const int sx = bx >> 5, sy = by >> 5;
    inside = (unsigned)sx < (unsigned)(PU.w - 1) && (unsigned)sy < (unsigned)(PU.h - 1);
    if (INTERLEAVED_UV) {
        typedef HotTap<T, true> Tap;
        uint32_t off = inside ? (uint32_t)sy * (uint32_t)PU.src_stride + (uint32_t)sx * (uint32_t)(2 * sizeof(T)) : 0u;
        if (!range_ok(aud, off, Tap::BYTES, PU.src_len) || !range_ok(aud, (int64_t)off + PU.src_stride, Tap::BYTES, PU.src_len)) off = 0u;
        const auto r0 = Tap::load(PU.src, off), r1 = Tap::load(PU.src, off + (uint32_t)PU.src_stride);
        const uint32_t w = Tap::wpack((uint32_t)bx & 31u);
        uint32_t u0, v0, u1, v1;
        Tap::dot(r0, w, u0, v0); Tap::dot(r1, w, u1, v1);
        ou = hot_blend(u0, u1, (uint32_t)by & 31u, lim_u);
        ov = hot_blend(v0, v1, (uint32_t)by & 31u, lim_u);
    } else {
        typedef HotTap<T, false> Tap;
        uint32_t off = inside ? (uint32_t)sy * (uint32_t)PU.src_stride + (uint32_t)sx * (uint32_t)sizeof(T) : 0u;
        const int top = PU.src_len < PV.src_len ? PU.src_len : PV.src_len;
        if (!range_ok(aud, off, Tap::BYTES, top) || !range_ok(aud, (int64_t)off + PU.src_stride, Tap::BYTES, top)) off = 0u;
        const uint32_t a0 = Tap::load(PU.src, off), a1 = Tap::load(PU.src, off + (uint32_t)PU.src_stride);
        const uint32_t b0 = Tap::load(PV.src, off), b1 = Tap::load(PV.src, off + (uint32_t)PU.src_stride);
        const uint32_t w = Tap::wpack((uint32_t)bx & 31u);
        ou = hot_blend(Tap::dot(a0, w), Tap::dot(a1, w), (uint32_t)by & 31u, lim_u);
        ov = hot_blend(Tap::dot(b0, w), Tap::dot(b1, w), (uint32_t)by & 31u, lim_v);
    }
}

// The kernel-argument segment as scalar-addressable constant memory: rare paths read their uniforms from here at the point
// of use instead of keeping them in (scarce) scalar registers across the pixel loop.
#ifndef GFW_HOT_ABLATE
#define GFW_HOT_ABLATE 0          // timing experiments only (wrong output): 1 no first pass, 2 no taps, 4 no exact resolve, 8 accept everything, 64 no luma store
#endif
typedef const GfwYuvArgs __attribute__((address_space(4))) *HotKArgs;
#define GFW_OPAQUE(p) asm volatile("" : "+s"(p))

template <typename T, int DH, bool INTERLEAVED_UV, bool AUDIT>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(GFW_WAVES_PER_EU, 8))) void gfw_hot_kernel(const GfwYuvArgs A) {
    constexpr int MODEL = GFW_MODEL_OPENCV_FISHEYE, DW = 2, RB = GFW_YUV_RB_FAST;
    constexpr int NPX = DW * DH;
    constexpr unsigned QCAP = 256;                 // ring of deferred pixels per wave: <= 63 pending + <= 128 new per step
    static_assert(RB * NPX <= 64, "slot index must fit 6 bits");
    __shared__ unsigned q_id[4][QCAP];             // (tile << 12) | (lane << 6) | (r * NPX + k)
    __shared__ unsigned short q_sy[4][QCAP];       // certified rolling-shutter row, or 0xFFFF: the exact first pass decides
    __shared__ unsigned q_tail[4];
    __shared__ unsigned short s_sy[RB * DH][2][256];   // phase 1 -> phase 2: certified row of each pixel, 0xFFFF = not certified (lane-private slots)
    const int wave = threadIdx.y, lane = threadIdx.x, tid = wave * 64 + lane;
    const bool two_pass = A.matrix_count > 1;
    const bool hrs = A.hrs != 0;
    unsigned long long *const aud = AUDIT ? A.audit : nullptr;
    HotKArgs Ak = (HotKArgs)__builtin_amdgcn_kernarg_segment_ptr();

    const float t2x = A.t2[0], t2y = A.t2[1], rl2 = A.r_limit_sq;
    const float bg_y = A.pl[0].bg[0], lim_y = A.pl[0].limit;
    const float bg_u = A.pl[1].bg[0], lim_u = A.pl[1].limit;
    const float bg_v = INTERLEAVED_UV ? A.pl[1].bg[1] : A.pl[2].bg[0], lim_v = INTERLEAVED_UV ? A.pl[1].limit : A.pl[2].limit;
    const GfwYuvPlane &PY = A.pl[0], &PU = A.pl[1], &PV = A.pl[INTERLEAVED_UV ? 1 : 2];
    const HotQ Q2{A.p1_rho_max, A.p1_rho_scale, A.p2_kappa};
    const HotC K{gfw_f2{A.f[0], A.f[1]}, gfw_f2{A.c[0], A.c[1]}, gfw_f2{A.map_lx.mul, A.map_ly.mul}, gfw_f2{A.map_cx.mul, A.map_cy.mul},
                 gfw_f2{-A.map_lx.den, -A.map_ly.den}, gfw_f2{A.map_lx.rcp, A.map_ly.rcp}};
    constexpr float top_f = sizeof(T) == 1 ? 255.0f : 65535.0f;
    const int row_lim = hrs ? A.width : A.height;
    Mid M{0, 0, 0, 0, 0, 0, 0, 0, 0};
    P1 Q1{0, 0, 0, 0, 0, 0};
    if (two_pass) {
        const float *mid = A.matrices + (size_t)(A.matrix_count >> 1) * GFW_MAT_STRIDE;       // wave-uniform -> scalar loads
        M = Mid{mid[0], mid[1], mid[2], mid[3], mid[4], mid[5], mid[6], mid[7], mid[8]};
        Q1 = P1{A.p1_rho_max, A.p1_rho_scale, A.p1_eps, A.p1_f, A.p1_c, (float)row_lim};
    }

    // ---- the exact path: `n` deferred pixels starting at ring position `head`, one per lane ------------------------------
    auto resolve = [&](unsigned head, unsigned n) {
        if ((unsigned)lane >= n) return;
        const unsigned idx = (head + (unsigned)lane) & (QCAP - 1u);
        const unsigned id = q_id[wave][idx];
        int sy = (int)q_sy[wave][idx];
        HotKArgs P = Ak;
        GFW_OPAQUE(P);                                           // uniforms of this path are read here, not carried through the pixel loop
        const int t = (int)(id >> 12), ql = (int)((id >> 6) & 63u), slot = (int)(id & 63u);
        const int r = slot / NPX, k = slot - r * NPX, i = k % DW, j = k / DW;
        const int tiles_x = P->tiles_x;
        const int ty = t / tiles_x, tx = t - ty * tiles_x;
        const int cx = tx * 64 + ql, cy = (ty * 4 + wave) * RB + r;
        const int lx = cx * DW + i, ly = cy * DH + j;
        const float ox = (float)lx + t2x, oy = (float)ly + t2y;
        Lens L;
        L.f0 = P->f[0]; L.f1 = P->f[1]; L.c0 = P->c[0]; L.c1 = P->c[1];
        L.k0 = P->k[0]; L.k1 = P->k[1]; L.k2 = P->k[2]; L.k3 = P->k[3];
        L.t2x = t2x; L.t2y = t2y; L.rl2 = rl2;
        const float *matrices = P->matrices;
        const int mc = P->matrix_count;
        if (sy == 0xFFFF) {                                      // the first pass was not certified either: cpu_undistort.rs:465-479
            const float *mid = matrices + (size_t)(mc >> 1) * GFW_MAT_STRIDE;
            const Mid M{mid[0], mid[1], mid[2], mid[3], mid[4], mid[5], mid[6], mid[7], mid[8]};
            sy = max(min(round_i32(hrs ? ox : oy), row_lim), 0);
            const float X = (ox * M.m0) + (oy * M.m1) + M.m2, Y = (ox * M.m3) + (oy * M.m4) + M.m5, W = (ox * M.m6) + (oy * M.m7) + M.m8;
            if (W > 0.0f && !(rl2 > 0.0f && (X * X + Y * Y) > rl2 * W)) {
                float u, v;
                const float mag = fmaxf(fmaxf(fabsf(X), fabsf(Y)), W);
                if (__builtin_expect((mag <= 524288.0f) && (W >= 9.5367431640625e-07f), 1)) fisheye_project<LeanOps>(X, Y, W, L, false, u, v);
                else fisheye_project<IeeeOps>(X, Y, W, L, false, u, v);
                sy = max(min(round_i32(hrs ? u : v), row_lim), 0);
            }
        }
        const float *m = matrices + (size_t)min(sy, mc - 1) * GFW_MAT_STRIDE;
        const float4 ma = *reinterpret_cast<const float4 *>(m), mb = *reinterpret_cast<const float4 *>(m + 4);
        const float X = (ox * ma.x) + (oy * ma.y) + ma.z, Y = (ox * ma.w) + (oy * mb.x) + mb.y, W = (ox * mb.z) + (oy * mb.w) + m[8];
        bool ok = W > 0.0f;
        if (rl2 > 0.0f && (X * X + Y * Y) > rl2 * W) ok = false;
        float u = 0.0f, v = 0.0f;
        if (ok) {
            const float mag = fmaxf(fmaxf(fabsf(X), fabsf(Y)), W);
            if (__builtin_expect((mag <= 524288.0f) && (W >= 9.5367431640625e-07f), 1)) fisheye_project<LeanOps>(X, Y, W, L, false, u, v);
            else fisheye_project<IeeeOps>(X, Y, W, L, false, u, v);
        }
        const gfw_f2 uv = {u, v};
        int bx, by;
        hot_bins(hot_map2(uv, K.mul_l, K.nden, K.rcp), bx, by);
        const uint32_t val = ok ? hot_sample<T>(PY, bx, by, bg_y, lim_y, aud) : gfw_f2u_sat(bg_y, top_f);
        const uint32_t doff = (uint32_t)ly * (uint32_t)PY.dst_stride + (uint32_t)lx * (uint32_t)sizeof(T);
        if (range_ok(aud, doff, sizeof(T), PY.dst_len)) *reinterpret_cast<T *>(PY.dst + doff) = (T)val;
        if (k == 0) {
            uint32_t ou = gfw_f2u_sat(bg_u, top_f), ov = gfw_f2u_sat(bg_v, top_f);
            if (ok) {
                int cbx, cby;
                hot_bins(hot_map2(uv, K.mul_c, K.nden, K.rcp), cbx, cby);
                hot_sample_uv<T, INTERLEAVED_UV>(PU, PV, cbx, cby, bg_u, bg_v, lim_u, lim_v, ou, ov, aud);
            }
            hot_store_uv<T, INTERLEAVED_UV>(PU, PV, cx, cy, ou, ov, aud);
        }
    };

    if (lane == 0) q_tail[wave] = 0;
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    unsigned head = 0;                                           // wave-uniform: entries [head, tail) of the ring are pending

    const int n_tiles = A.tiles_x * A.tiles_y;
    const int per_xcd = (n_tiles + 7) >> 3;
    const int wg_per_xcd = (int)gridDim.x >> 3;
    const int xcd = (int)blockIdx.x & 7;
    for (int tb = (int)blockIdx.x >> 3; tb < per_xcd; tb += wg_per_xcd) {
        const int t = xcd * per_xcd + tb;
        if (t >= n_tiles) break;
        const int ty = t / A.tiles_x, tx = t - ty * A.tiles_x;
        const int cx = tx * 64 + lane;
        const int cy0 = (ty * 4 + wave) * RB;
        const bool lane_ok = cx < A.cw;
        const int lx0 = cx * DW;
        const gfw_f2 oxp = {(float)lx0 + t2x, (float)(lx0 + 1) + t2x};

        // ---- phase 1: certified rolling-shutter rows of the lane's RB x DH pixel pairs, straight-line: the table lookups of all
        // rows are in flight together.  Rows outside the frame are computed too (harmless) and never used.
        if (two_pass && !(GFW_HOT_ABLATE & 1)) {
            #pragma unroll
            for (int rj = 0; rj < RB * DH; ++rj) {
                const int ly = cy0 * DH + rj;
                const float oy = (float)ly + t2y;
                int sy0, sy1; bool g0, g1; gfw_f2 v_fast;
                hot_pass1_pair(oxp, oy, M, Q1, A.p1_table, hrs, rl2, sy0, sy1, g0, g1, v_fast, aud);
                s_sy[rj][0][tid] = g0 ? (unsigned short)sy0 : (unsigned short)0xFFFF;
                s_sy[rj][1][tid] = g1 ? (unsigned short)sy1 : (unsigned short)0xFFFF;
                if (AUDIT && lane_ok && ly < A.out_h) {
                    Lens L;
                    L.f0 = A.f[0]; L.f1 = A.f[1]; L.c0 = A.c[0]; L.c1 = A.c[1]; L.k0 = A.k[0]; L.k1 = A.k[1]; L.k2 = A.k[2]; L.k3 = A.k[3];
                    L.t2x = t2x; L.t2y = t2y; L.rl2 = rl2;
                    #pragma unroll
                    for (int i = 0; i < DW; ++i) {
                        if (lx0 + i >= A.out_w) continue;
                        if (!(i ? g1 : g0)) { atomicAdd(&A.audit[2], 1ull); continue; }
                        atomicAdd(&A.audit[0], 1ull);
                        const float oxi = i ? oxp.y : oxp.x;
                        if (pass1_exact<MODEL>(oxi, oy, M, L, A) != (i ? sy1 : sy0)) atomicAdd(&A.audit[1], 1ull);
                        const GfwPt ex = rd<MODEL>(oxi, oy, float4{M.m0, M.m1, M.m2, M.m3}, float4{M.m4, M.m5, M.m6, M.m7}, M.m8, A.matrices + (size_t)(A.matrix_count / 2) * GFW_MAT_STRIDE + 8, L, A);
                        if (ex.ok) atomicMax(&A.audit[4], (unsigned long long)gfw_f2u(fabsf((hrs ? ex.x : ex.y) - (i ? v_fast.y : v_fast.x))));
                    }
                }
            }
        }

        #pragma unroll 1
        for (int rj = 0; rj < RB * DH; ++rj) {
            const int r = rj / DH, j = rj - r * DH;
            const int cy = cy0 + r, ly = cy * DH + j;
            if (lane_ok && cy < A.ch && ly < A.out_h) {
                const float oy = (float)ly + t2y;
                int sy[2]; bool good[2] = {true, true};
                if (two_pass && !(GFW_HOT_ABLATE & 1)) {
                    sy[0] = (int)s_sy[rj][0][tid]; sy[1] = (int)s_sy[rj][1][tid];
                    good[0] = sy[0] != 0xFFFF; good[1] = sy[1] != 0xFFFF;
                } else {
                    sy[0] = max(min(round_i32(hrs ? oxp.x : oy), row_lim), 0);
                    sy[1] = max(min(round_i32(hrs ? oxp.y : oy), row_lim), 0);
                }
                // ---- second pass + taps of the two pixels together; what is not certified is deferred to the exact path ----------
                const bool with_chroma = (j == 0);
                const bool px1 = lx0 + 1 < A.out_w;                  // odd output widths: the pair's second pixel may not exist
                const int row0 = min(sy[0], A.matrix_count - 1), row1 = min(sy[1], A.matrix_count - 1);
                if (AUDIT && ((unsigned)row0 >= (unsigned)A.matrix_count || (unsigned)row1 >= (unsigned)A.matrix_count)) atomicAdd(&A.audit[5], 1ull);
                HotPair R;
                hot_project_pair(oxp, oy, A.matrices, (uint32_t)row0 * (uint32_t)(GFW_MAT_STRIDE * sizeof(float)),
                                 (uint32_t)row1 * (uint32_t)(GFW_MAT_STRIDE * sizeof(float)), rl2, K, Q2, A.p1_table, with_chroma, R, aud);
                if (GFW_HOT_ABLATE & 8) { R.acc[0] = true; R.acc[1] = true; }
                // settled now: certified (or invalid: background colour, an exact decision); otherwise the exact path takes the pixel
                const bool done0 = good[0] & (R.acc[0] | !R.ok[0]);
                const bool done1 = px1 & good[1] & (R.acc[1] | !R.ok[1]);
                if (AUDIT) {                                         // audit: the certified bins against the exact projection's
                    Lens L;
                    L.f0 = A.f[0]; L.f1 = A.f[1]; L.c0 = A.c[0]; L.c1 = A.c[1]; L.k0 = A.k[0]; L.k1 = A.k[1]; L.k2 = A.k[2]; L.k3 = A.k[3];
                    L.t2x = t2x; L.t2y = t2y; L.rl2 = rl2;
                    #pragma unroll
                    for (int i = 0; i < DW; ++i) {
                        if (!(i ? done1 : done0) || !R.ok[i]) continue;
                        atomicAdd(&A.audit[6], 1ull);
                        const GfwPt p = rd_row<MODEL>(i ? oxp.y : oxp.x, oy, i ? row1 : row0, L, A);
                        int ex, ey;
                        hot_bins(hot_map2(gfw_f2{p.x, p.y}, K.mul_l, K.nden, K.rcp), ex, ey);
                        bool same = p.ok && ex == R.bx[i] && ey == R.by[i];
                        if (i == 0 && with_chroma) { hot_bins(hot_map2(gfw_f2{p.x, p.y}, K.mul_c, K.nden, K.rcp), ex, ey); same = same && ex == R.cbx && ey == R.cby; }
                        if (!same) atomicAdd(&A.audit[7], 1ull);
                    }
                }
                uint32_t val0, val1, ou = 0u, ov = 0u;
                bool in0, in1, inc = true;
                if (GFW_HOT_ABLATE & 2) { val0 = (uint32_t)(R.bx[0] + R.by[0] + R.cbx); val1 = (uint32_t)(R.bx[1] + R.by[1] + R.cby); in0 = in1 = true; }   // timing ablation only
                else {
                    val0 = hot_sample_free<T>(PY, A.width, A.height, R.bx[0], R.by[0], lim_y, in0, aud);
                    val1 = hot_sample_free<T>(PY, A.width, A.height, R.bx[1], R.by[1], lim_y, in1, aud);
                    if (with_chroma) hot_sample_uv_free<T, INTERLEAVED_UV>(PU, PV, R.cbx, R.cby, lim_u, lim_v, ou, ov, inc, aud);
                }
                // rare: taps that straddle the source rect (cpu_undistort.rs:392-409), for the lanes that need them
                if (__builtin_expect((done0 & R.ok[0] & !in0) | (done1 & R.ok[1] & !in1) | (with_chroma & done0 & R.ok[0] & !inc), 0)) {
                    if (done0 & R.ok[0] & !in0) val0 = hot_sample<T>(PY, R.bx[0], R.by[0], bg_y, lim_y, aud);
                    if (done1 & R.ok[1] & !in1) val1 = hot_sample<T>(PY, R.bx[1], R.by[1], bg_y, lim_y, aud);
                    if (with_chroma & done0 & R.ok[0] & !inc) hot_sample_uv<T, INTERLEAVED_UV>(PU, PV, R.cbx, R.cby, bg_u, bg_v, lim_u, lim_v, ou, ov, aud);
                }
                if (!R.ok[0]) { val0 = gfw_f2u_sat(bg_y, top_f); ou = gfw_f2u_sat(bg_u, top_f); ov = gfw_f2u_sat(bg_v, top_f); }
                if (!R.ok[1]) val1 = gfw_f2u_sat(bg_y, top_f);
                if (with_chroma & done0) hot_store_uv<T, INTERLEAVED_UV>(PU, PV, cx, cy, ou, ov, aud);
                // the lane's two horizontally adjacent luma pixels: one store when both are settled
                const uint32_t doff = (uint32_t)ly * (uint32_t)PY.dst_stride + (uint32_t)lx0 * (uint32_t)sizeof(T);
                if ((GFW_HOT_ABLATE & 64) && val0 + val1 != 0x12345u) { }
                else if (done0 & done1) {
                    if (range_ok(aud, doff, 2 * sizeof(T), PY.dst_len)) {
                        if (sizeof(T) == 2) *reinterpret_cast<uint32_t *>(PY.dst + doff) = val0 | (val1 << 16);
                        else *reinterpret_cast<uint16_t *>(PY.dst + doff) = (uint16_t)(val0 | (val1 << 8));
                    }
                } else {
                    if (done0 && range_ok(aud, doff, sizeof(T), PY.dst_len)) *reinterpret_cast<T *>(PY.dst + doff) = (T)val0;
                    if (done1 && range_ok(aud, (int64_t)doff + sizeof(T), sizeof(T), PY.dst_len)) *reinterpret_cast<T *>(PY.dst + doff + sizeof(T)) = (T)val1;
                }
                if (!done0 | (px1 & !done1)) {                       // the exact path decides (and samples) these pixels
                    #pragma unroll
                    for (int i = 0; i < DW; ++i) {
                        if (i ? (!px1 | done1) : done0) continue;
                        const unsigned slot = atomicAdd(&q_tail[wave], 1u) & (QCAP - 1u);
                        q_id[wave][slot] = ((unsigned)t << 12) | ((unsigned)lane << 6) | (unsigned)(r * NPX + j * DW + i);
                        q_sy[wave][slot] = good[i] ? (unsigned short)sy[i] : (unsigned short)0xFFFF;
                    }
                }
            }
            // drain the ring in full chunks of 64: the exact path always runs with every lane busy
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            const unsigned tail = q_tail[wave];
            while (tail - head >= 64u) { if (!(GFW_HOT_ABLATE & 4)) resolve(head, 64u); head += 64u; }
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    const unsigned tail = q_tail[wave];
    if (tail != head) resolve(head, tail - head);
}

template <typename T, bool AUDIT>
hipError_t launch_hot(const GfwYuvArgs &A, int dh, bool interleaved, hipStream_t s) {
    const int n_tiles = A.tiles_x * A.tiles_y;
    if (n_tiles <= 0) return hipSuccess;
    int grid = A.grid_limit > 0 ? A.grid_limit : 256 * 6;
    const int per_xcd = (n_tiles + 7) >> 3;
    if (grid > per_xcd * 8) grid = per_xcd * 8;
    grid = (grid + 7) & ~7;
    dim3 block(64, 4);
    if (dh == 1 && !interleaved) hipLaunchKernelGGL((gfw_hot_kernel<T, 1, false, AUDIT>), dim3(grid), block, 0, s, A);
    else if (dh == 1 && interleaved) hipLaunchKernelGGL((gfw_hot_kernel<T, 1, true, AUDIT>), dim3(grid), block, 0, s, A);
    else if (dh == 2 && !interleaved) hipLaunchKernelGGL((gfw_hot_kernel<T, 2, false, AUDIT>), dim3(grid), block, 0, s, A);
    else if (dh == 2 && interleaved) hipLaunchKernelGGL((gfw_hot_kernel<T, 2, true, AUDIT>), dim3(grid), block, 0, s, A);
    else return hipErrorInvalidValue;
    return hipGetLastError();
}
#endif   // hot kernel

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
#if GFW_HOT_ONLY
    if (dw == 2 && dh == 1 && !interleaved) GFW_YUV_LAUNCH(2, 1, false); else return hipErrorInvalidValue;
#else
    if constexpr (N0 > 1 || is_f32<T>::value) {          // packed single plane, or planar f32 planes: full resolution only
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
#endif
#undef GFW_YUV_LAUNCH
    return hipGetLastError();
}

}  // namespace

template <int MODEL, typename T, int N0>
static hipError_t launch_tn(const GfwYuvArgs &A, int dw, int dh, bool interleaved, bool fast1, hipStream_t s) {
    constexpr int I = GFW_FRAME_TAPS;
    if constexpr (MODEL == GFW_MODEL_OPENCV_FISHEYE) if (fast1) {      // the certified first pass exists for the specialised fisheye model only
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
#if GFW_HOT_ONLY
    if (A.model == GFW_MODEL_OPENCV_FISHEYE && !A.extras && n0 == 1 && dw == 2 && dh == 1 && !interleaved && fast1 && !A.audit && !A.hot && GFW_FRAME_KIND == 2 && GFW_FRAME_TAPS == 2) {
        const hipError_t e = launch_mt<GFW_MODEL_OPENCV_FISHEYE, uint16_t, 1, 2, GFW_YUV_RB_FAST, true, false>(A, dw, dh, interleaved, s);
#if GFW_TIMELINE
        static int n_launch = 0;
        if (++n_launch == 60 && getenv("GFW_TIMELINE_FILE")) {
            static unsigned long long host[8192 * 8];
            (void)hipStreamSynchronize(s);
            (void)hipMemcpyFromSymbol(host, HIP_SYMBOL(gfw_tl), sizeof(host));
            if (FILE *f = fopen(getenv("GFW_TIMELINE_FILE"), "wb")) { fwrite(host, 1, sizeof(host), f); fclose(f); }
        }
#endif
        return e;
    }
    return hipErrorInvalidValue;
#else
#if GFW_FRAME_TAPS == 2 && GFW_FRAME_KIND != 4 && !GFW_HOT_ONLY
    if (A.hot) {
        if (n0 != 1 || dw != 2 || A.model != GFW_MODEL_OPENCV_FISHEYE || A.extras) return hipErrorInvalidValue;
#if GFW_FRAME_KIND == 1
        return A.audit ? launch_hot<uint8_t, true>(A, dh, interleaved, s) : launch_hot<uint8_t, false>(A, dh, interleaved, s);
#else
        return A.audit ? launch_hot<uint16_t, true>(A, dh, interleaved, s) : launch_hot<uint16_t, false>(A, dh, interleaved, s);
#endif
    }
#endif
    if (A.model == GFW_MODEL_OPENCV_FISHEYE && !A.extras) return launch_m<GFW_MODEL_OPENCV_FISHEYE>(A, n0, dw, dh, interleaved, fast1, s);
    return launch_m<-1>(A, n0, dw, dh, interleaved, false, s);
#endif
}
