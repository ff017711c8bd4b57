/* gfw_oracle.c — CPU restatement of gyroflow-core's warp path.  TEST INFRASTRUCTURE ONLY.
 *
 * This file is the parity oracle for libgfwarp.  It is a plain-C restatement
 * (no code copied; Rust -> C by hand) of
 *
 *   src/core/stabilization/cpu_undistort.rs:133-228   Stabilization::rotate_and_distort
 *   src/core/stabilization/cpu_undistort.rs:233-633   Stabilization::undistort_image_cpu::<I,T>
  *   src/core/stabilization/distortion_models/<model>.rs    distort_point / undistort_point (14 models)
 *   src/core/stabilization/pixel_formats.rs           PixelType::to_float / from_float
 *   src/core/util.rs:144-147                          map_coord
 *   src/core/gyro_source/splines.rs:88-177, sony.rs:557-563   interpolate_mesh
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it.
 * The product (libgfwarp) never links, loads or calls anything in oracle/.
 *
 * PARITY STATUS: *pinned on the reference's own code* since round 3 — gyroflow ships no test, golden vector or
 * fixture for this path and its Rust cannot be built here (no cargo/rustc), but its OpenCL
 * twin of the path can be: oracle/build_ref_cl.py assembles src/core/gpu/opencl_undistort.cl + the lens model's
 * .cl the way OclWrapper::new does (opencl.rs:181-214) from /root/reference and compiles that text for the host
 * cores; oracle/ref_cl_host.c supplies the OpenCL builtins (transcendentals from glibc — what the CPU path calls).
 *   - tests/golden/ref_golden.json holds what THAT code wrote for 44 configurations in which none of the twin's
 *     documented deviations from the CPU path can fire (BASELINE's C2 frame at 3840x2160, C1 1080p, C4's crop, every
 *     pixel type its OpenCL backend serves x bilinear / bicubic / Lanczos4, both shutter directions, edge-repeat and
 *     mirror backgrounds, stretches, rescaled output, four more lens models, the five digital lenses, IBIS terms, quarter-turn input rotation): this file reproduces every plane BIT
 *     FOR BIT (tests/test_ref_golden.py, CPU tier), and so does libgfwarp (GPU tier);
 *   - where the deviations do fire (negative coordinates under the twin's rtz rounding, the r-limit formula, NaN
 *     coordinates, background mode 3's feather zone) every differing pixel is attributed to one of them, zero
 *     unexplained, at a 2e-5 px tolerance (tests/test_ref_opencl_host.py); EWA agrees to one code value;
 *   - the same twin compiled for gfx950 runs beside this file on the GPU box (tests/test_gpu_ref_opencl.py; there its
 *     atan / tan are OpenCL's, hence a 2e-4 px tolerance);
 *   - what no twin can pin — the CPU-only colour-range fix (cpu_undistort.rs:255-260 differs from .cl:157-160 by
 *     design), the Sony mesh beyond 99.9 % (f64 spline here, f32 in the twin), three-channel pixels — rests on: analytic known-answer tests (tests/test_oracle_kat.py), 40-digit mpmath statements of the 14
 *     lens models in both directions written from the Rust independently of this file (tests/test_oracle_mpmath.py),
 *     and self-generated golden checksums (tests/golden/golden.json).
 *
 * Rust semantics honoured here:
 *   f32::round      = half away from zero            -> roundf
 *   `as i32/u8/u16` = truncate, saturate, NaN -> 0   -> f2i / f2u8 / f2u16
 *   f32::max/min    = IEEE maxNum/minNum             -> fmaxf / fminf
 *   powi(2)         = x*x
 *   no FMA contraction (build with -ffp-contract=off), IEEE div/sqrt,
 *   atan/tan/sin/cos = this box's libm (glibc 2.35), exactly what the Rust
 *   binary would call through std on linux-gnu.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <stdio.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#include "../include/gfwarp.h"

typedef gfw_kernel_params KP;
typedef struct { float x, y, z, w; } v4;
typedef struct { int ok; float x, y; } opt2;

/* ------------------------------------------------------------------ casts */
static inline int32_t f2i(float v) {            /* Rust `v as i32` */
    if (v != v) return 0;
    if (v >= 2147483648.0f) return INT32_MAX;
    if (v <= -2147483648.0f) return INT32_MIN;
    return (int32_t)v;
}
static inline uint8_t f2u8(float v) {           /* Rust `v as u8` */
    if (v != v) return 0;
    if (v >= 255.0f) return 255;
    if (v <= 0.0f) return 0;
    return (uint8_t)v;
}
static inline uint16_t f2u16(float v) {         /* Rust `v as u16` */
    if (v != v) return 0;
    if (v >= 65535.0f) return 65535;
    if (v <= 0.0f) return 0;
    return (uint16_t)v;
}
static inline int64_t d2usize(double v) {       /* Rust `v as usize` (f64) */
    if (v != v) return 0;
    if (v <= 0.0) return 0;
    if (v >= 9.2e18) return INT64_MAX;
    return (int64_t)v;
}
static inline float rs_clamp(float x, float lo, float hi) { /* f32::clamp */
    if (x < lo) x = lo;
    if (x > hi) x = hi;
    return x;
}
/* half crate 2.7.1 f16::from_f32 / to_f32 (pixel_formats.rs:239-243): IEEE RNE */
static uint16_t f32_to_f16(float f) {
    uint32_t x; memcpy(&x, &f, 4);
    uint32_t sign = (x >> 16) & 0x8000u;
    uint32_t man = x & 0x007FFFFFu;
    int32_t exp = (int32_t)((x >> 23) & 0xFF);
    if (exp == 255) {                                   /* inf / nan */
        uint32_t nan_bit = man ? 0x0200u : 0;
        return (uint16_t)(sign | 0x7C00u | nan_bit | (man >> 13));
    }
    int32_t unbiased = exp - 127;
    int32_t half_exp = unbiased + 15;
    if (half_exp >= 31) return (uint16_t)(sign | 0x7C00u);     /* overflow -> inf */
    if (half_exp <= 0) {                                /* subnormal or zero */
        if (14 - half_exp > 24) return (uint16_t)sign;
        man |= 0x00800000u;
        uint32_t shift = (uint32_t)(14 - half_exp);
        uint32_t half_man = man >> shift;
        uint32_t round_bit = 1u << (shift - 1);
        if ((man & round_bit) != 0 && (man & (3 * round_bit - 1)) != 0) half_man += 1;
        return (uint16_t)(sign | half_man);
    }
    uint32_t half_man = man >> 13;
    uint32_t out = sign | ((uint32_t)half_exp << 10) | half_man;
    uint32_t round_bit = 0x00001000u;
    if ((man & round_bit) != 0 && (man & (3 * round_bit - 1)) != 0) out += 1;
    return (uint16_t)out;
}
static float f16_to_f32(uint16_t h) {
    uint32_t sign = ((uint32_t)h & 0x8000u) << 16;
    uint32_t exp = (h >> 10) & 0x1F;
    uint32_t man = h & 0x3FFu;
    uint32_t out;
    if (exp == 0) {
        if (man == 0) out = sign;
        else {
            int e = -1;
            do { e++; man <<= 1; } while ((man & 0x400u) == 0);
            out = sign | ((uint32_t)(127 - 15 - e) << 23) | ((man & 0x3FFu) << 13);
        }
    } else if (exp == 31) {
        out = sign | 0x7F800000u | (man << 13);
    } else {
        out = sign | ((exp + 112) << 23) | (man << 13);
    }
    float f; memcpy(&f, &out, 4); return f;
}

/* util.rs:144-147 */
static inline float map_coord(float x, float in_min, float in_max, float out_min, float out_max) {
    return (x - in_min) * (out_max - out_min) / (in_max - in_min) + out_min;
}

/* cpu_undistort.rs:11-58: the tap LUT.  Bilinear phases are k/32 exactly; the
 * bicubic / Lanczos4 rows are the reference's printed 6-decimal literals. */
#include "gfw_coeffs.inc"

/* ------------------------------------------------- distortion models -----
 * Each pair follows distortion_models/<name>.rs; line refs in comments.  */

/* opencv_fisheye.rs:12-70 / :72-95 */
static opt2 fisheye_undistort(float px, float py, const KP *p) {
    opt2 r = {1, px, py};
    if (p->k[0] == 0.0f && p->k[1] == 0.0f && p->k[2] == 0.0f && p->k[3] == 0.0f) return r;
    const float EPS = 1e-6f;
    float theta_d = sqrtf(px * px + py * py);
    theta_d = fminf(fmaxf(theta_d, -3.14159265358979323846f), 3.14159265358979323846f);
    int converged = 0;
    float theta = theta_d;
    float scale = 0.0f;
    if (fabsf(theta_d) > EPS) {
        theta = 0.0f;
        for (int i = 0; i < 10; ++i) {
            float theta2 = theta * theta;
            float theta4 = theta2 * theta2;
            float theta6 = theta4 * theta2;
            float theta8 = theta6 * theta2;
            float k0_theta2 = p->k[0] * theta2;
            float k1_theta4 = p->k[1] * theta4;
            float k2_theta6 = p->k[2] * theta6;
            float k3_theta8 = p->k[3] * theta8;
            float theta_fix = (theta * (1.0f + k0_theta2 + k1_theta4 + k2_theta6 + k3_theta8) - theta_d)
                            / (1.0f + 3.0f * k0_theta2 + 5.0f * k1_theta4 + 7.0f * k2_theta6 + 9.0f * k3_theta8);
            theta_fix = fminf(fmaxf(theta_fix, -0.9f), 0.9f);
            theta = theta - theta_fix;
            if (fabsf(theta_fix) < EPS) { converged = 1; break; }
        }
        scale = tanf(theta) / theta_d;
    } else {
        converged = 1;
    }
    int theta_flipped = (theta_d < 0.0f && theta > 0.0f) || (theta_d > 0.0f && theta < 0.0f);
    if (converged && !theta_flipped) { r.x = px * scale; r.y = py * scale; return r; }
    r.ok = 0; return r;
}
static void fisheye_distort(float x, float y, float z, const KP *p, float *ox, float *oy) {
    x = x / z; y = y / z;
    if (p->k[0] == 0.0f && p->k[1] == 0.0f && p->k[2] == 0.0f && p->k[3] == 0.0f) { *ox = x; *oy = y; return; }
    float r = sqrtf(x * x + y * y);
    float theta = atanf(r);
    float theta2 = theta * theta;
    float theta4 = theta2 * theta2;
    float theta6 = theta4 * theta2;
    float theta8 = theta4 * theta4;
    float theta_d = theta * (1.0f + p->k[0] * theta2 + p->k[1] * theta4 + p->k[2] * theta6 + p->k[3] * theta8);
    float scale = (r == 0.0f) ? 1.0f : theta_d / r;
    *ox = x * scale; *oy = y * scale;
}

/* opencv_standard.rs:12-31 / :33-49 */
static opt2 cvstd_undistort(float px, float py, const KP *p) {
    opt2 r = {1, 0, 0};
    float x = px, y = py, x0 = px, y0 = py;
    const float *k = p->k;
    for (int i = 0; i < 20; ++i) {
        float r2 = x * x + y * y;
        float icdist = (1.0f + ((k[7] * r2 + k[6]) * r2 + k[5]) * r2) / (1.0f + ((k[4] * r2 + k[1]) * r2 + k[0]) * r2);
        if (icdist < 0.0f) { r.ok = 0; return r; }
        float delta_x = 2.0f * k[2] * x * y + k[3] * (r2 + 2.0f * x * x) + k[8]  * r2 + k[9]  * r2 * r2;
        float delta_y = k[2] * (r2 + 2.0f * y * y) + 2.0f * k[3] * x * y + k[10] * r2 + k[11] * r2 * r2;
        x = (x0 - delta_x) * icdist;
        y = (y0 - delta_y) * icdist;
    }
    r.x = x; r.y = y; return r;
}
static void cvstd_distort(float x, float y, float z, const KP *p, float *ox, float *oy) {
    const float *k = p->k;
    x = x / z; y = y / z;
    float r2 = x * x + y * y;
    float r4 = r2 * r2;
    float r6 = r4 * r2;
    float a1 = 2.0f * x * y;
    float a2 = r2 + 2.0f * x * x;
    float a3 = r2 + 2.0f * y * y;
    float cdist = 1.0f + k[0] * r2 + k[1] * r4 + k[4] * r6;
    float icdist2 = 1.0f / (1.0f + k[5] * r2 + k[6] * r4 + k[7] * r6);
    *ox = x * cdist * icdist2 + k[2] * a1 + k[3] * a2 + k[8]  * r2 + k[9]  * r4;
    *oy = y * cdist * icdist2 + k[2] * a3 + k[3] * a1 + k[10] * r2 + k[11] * r4;
}

/* poly3.rs:15-51 / :53-62 */
#define NEWTON_EPS 0.00001f
static opt2 poly3_undistort(float px, float py, const KP *p) {
    opt2 r = {0, 0, 0};
    float inv_k1 = 1.0f / p->k[0];
    float rd = sqrtf(px * px + py * py);
    if (rd == 0.0f) return r;
    float rd_div_k1 = rd * inv_k1;
    float ru = rd;
    for (int i = 0; i < 10; ++i) {
        float fru = ru * ru * ru + ru * inv_k1 - rd_div_k1;
        if (fru >= -NEWTON_EPS && fru < NEWTON_EPS) break;
        if (i > 5) return r;
        ru = ru - (fru / (3.0f * ru * ru + inv_k1));
    }
    if (ru < 0.0f) return r;
    ru = ru / rd;
    r.ok = 1; r.x = px * ru; r.y = py * ru; return r;
}
static void poly3_distort(float x, float y, float z, const KP *p, float *ox, float *oy) {
    x = x / z; y = y / z;
    float poly2 = p->k[0] * (x * x + y * y) + 1.0f;
    *ox = x * poly2; *oy = y * poly2;
}

/* poly5.rs:13-42 / :44-54 */
static opt2 poly5_undistort(float px, float py, const KP *p) {
    opt2 r = {0, 0, 0};
    float rd = sqrtf(px * px + py * py);
    if (rd == 0.0f) return r;
    float ru = rd;
    for (int i = 0; i < 10; ++i) {
        float ru2 = ru * ru;
        float fru = ru * (1.0f + p->k[0] * ru2 + p->k[1] * ru2 * ru2) - rd;
        if (fru >= -NEWTON_EPS && fru < NEWTON_EPS) break;
        if (i > 5) return r;
        ru = ru - (fru / (1.0f + 3.0f * p->k[0] * ru2 + 5.0f * p->k[1] * ru2 * ru2));
    }
    if (ru < 0.0f) return r;
    ru = ru / rd;
    r.ok = 1; r.x = px * ru; r.y = py * ru; return r;
}
static void poly5_distort(float x, float y, float z, const KP *p, float *ox, float *oy) {
    x = x / z; y = y / z;
    float ru2 = x * x + y * y;
    float poly4 = 1.0f + p->k[0] * ru2 + p->k[1] * ru2 * ru2;
    *ox = x * poly4; *oy = y * poly4;
}

/* ptlens.rs:13-41 / :43-54 */
static opt2 ptlens_undistort(float px, float py, const KP *p) {
    opt2 r = {0, 0, 0};
    const float *k = p->k;
    float rd = sqrtf(px * px + py * py);
    if (rd == 0.0f) return r;
    float ru = rd;
    for (int i = 0; i < 10; ++i) {
        float fru = ru * (k[0] * ru * ru * ru + k[1] * ru * ru + k[2] * ru + 1.0f) - rd;
        if (fru >= -NEWTON_EPS && fru < NEWTON_EPS) break;
        if (i > 5) return r;
        ru = ru - (fru / (4.0f * k[0] * ru * ru * ru + 3.0f * k[1] * ru * ru + 2.0f * k[2] * ru + 1.0f));
    }
    if (ru < 0.0f) return r;
    ru = ru / rd;
    r.ok = 1; r.x = px * ru; r.y = py * ru; return r;
}
static void ptlens_distort(float x, float y, float z, const KP *p, float *ox, float *oy) {
    const float *k = p->k;
    x = x / z; y = y / z;
    float ru2 = x * x + y * y;
    float r = sqrtf(ru2);
    float poly3 = k[0] * ru2 * r + k[1] * ru2 + k[2] * r + 1.0f;
    *ox = x * poly3; *oy = y * poly3;
}

/* insta360.rs:27-50 / :10-25 */
static void insta360_distort(float x, float y, float z, const KP *p, float *ox, float *oy) {
    float k1 = p->k[0], k2 = p->k[1], k3 = p->k[2], p1 = p->k[3], p2 = p->k[4], xi = p->k[5];
    float len = sqrtf(x * x + y * y + z * z);
    x = (x / len) / ((z / len) + xi);
    y = (y / len) / ((z / len) + xi);
    float r2 = x * x + y * y;
    float r4 = r2 * r2;
    float r6 = r4 * r2;
    *ox = x * (1.0f + k1 * r2 + k2 * r4 + k3 * r6) + 2.0f * p1 * x * y + p2 * (r2 + 2.0f * x * x);
    *oy = y * (1.0f + k1 * r2 + k2 * r4 + k3 * r6) + 2.0f * p2 * x * y + p1 * (r2 + 2.0f * y * y);
}
static opt2 insta360_undistort(float ptx, float pty, const KP *p) {
    opt2 r = {1, 0, 0};
    float px = ptx, py = pty;
    for (int i = 0; i < 200; ++i) {
        float dx, dy;
        insta360_distort(px, py, 1.0f, p, &dx, &dy);
        float d0 = dx - ptx, d1 = dy - pty;
        if (fabsf(d0) < 1e-6f && fabsf(d1) < 1e-6f) break;
        px -= d0; py -= d1;
    }
    r.x = px; r.y = py; return r;
}

/* sony.rs:10-63 / :65-89 */
static opt2 sony_undistort(float px, float py, const KP *p) {
    opt2 r = {1, px, py};
    const float *k = p->k;
    if (k[0] == 0.0f && k[1] == 0.0f && k[2] == 0.0f && k[3] == 0.0f) return r;
    const float EPS = 1e-6f;
    float theta_d = sqrtf(px * px + py * py);
    int converged = 0;
    float theta = theta_d;
    float scale = 0.0f;
    if (fabsf(theta_d) > EPS) {
        theta = 0.0f;
        for (int i = 0; i < 10; ++i) {
            float theta2 = theta * theta;
            float theta3 = theta2 * theta;
            float theta4 = theta2 * theta2;
            float theta5 = theta2 * theta3;
            float k0 = k[0];
            float k1_theta1 = k[1] * theta;
            float k2_theta2 = k[2] * theta2;
            float k3_theta3 = k[3] * theta3;
            float k4_theta4 = k[4] * theta4;
            float k5_theta5 = k[5] * theta5;
            float theta_fix = (theta * (k0 + k1_theta1 + k2_theta2 + k3_theta3 + k4_theta4 + k5_theta5) - theta_d)
                            / (k0 + 2.0f * k1_theta1 + 3.0f * k2_theta2 + 4.0f * k3_theta3 + 5.0f * k4_theta4 + 6.0f * k5_theta5);
            theta = theta - theta_fix;
            if (fabsf(theta_fix) < EPS) { converged = 1; break; }
        }
        scale = tanf(theta) / theta_d;
    } else {
        converged = 1;
    }
    int theta_flipped = (theta_d < 0.0f && theta > 0.0f) || (theta_d > 0.0f && theta < 0.0f);
    if (converged && !theta_flipped) { r.x = px * scale; r.y = py * scale; return r; }
    r.ok = 0; return r;
}
static void sony_distort(float x, float y, float z, const KP *p, float *ox, float *oy) {
    const float *k = p->k;
    x = x / z; y = y / z;
    if (k[0] == 0.0f && k[1] == 0.0f && k[2] == 0.0f && k[3] == 0.0f) { *ox = x; *oy = y; return; }
    float r = sqrtf(x * x + y * y);
    float theta = atanf(r);
    float theta2 = theta * theta;
    float theta3 = theta2 * theta;
    float theta4 = theta2 * theta2;
    float theta5 = theta2 * theta3;
    float theta6 = theta3 * theta3;
    float theta_d = theta * k[0] + theta2 * k[1] + theta3 * k[2] + theta4 * k[3] + theta5 * k[4] + theta6 * k[5];
    float scale = (r == 0.0f) ? 1.0f : theta_d / r;
    *ox = x * scale; *oy = y * scale;
}

/* generic_polynomial.rs:10-78 / :80-120 */
static int genpoly_all_zero(const float *k) {
    for (int i = 0; i < 12; ++i) if (!(k[i] == 0.0f)) return 0;
    return 1;
}
static opt2 genpoly_undistort(float px, float py, const KP *p) {
    opt2 r = {1, px, py};
    const float *k = p->k;
    if (genpoly_all_zero(k)) return r;
    const float EPS = 1e-6f;
    float theta_d = sqrtf(px * px + py * py);
    int converged = 0;
    float theta = theta_d;
    float scale = 0.0f;
    if (fabsf(theta_d) > EPS) {
        theta = 0.0f;
        for (int i = 0; i < 10; ++i) {
            float theta2  = theta * theta;
            float theta3  = theta2 * theta;
            float theta4  = theta2 * theta2;
            float theta5  = theta2 * theta3;
            float theta6  = theta3 * theta3;
            float theta7  = theta3 * theta4;
            float theta8  = theta4 * theta4;
            float theta9  = theta4 * theta5;
            float theta10 = theta5 * theta5;
            float theta11 = theta5 * theta6;
            float k0 = k[0];
            float k1t = k[1] * theta,  k2t = k[2] * theta2, k3t = k[3] * theta3, k4t = k[4] * theta4;
            float k5t = k[5] * theta5, k6t = k[6] * theta6, k7t = k[7] * theta7, k8t = k[8] * theta8;
            float k9t = k[9] * theta9, k10t = k[10] * theta10, k11t = k[11] * theta11;
            float theta_fix = (theta * (k0 + k1t + k2t + k3t + k4t + k5t + k6t + k7t + k8t + k9t + k10t + k11t) - theta_d)
                            / (k0 + 2.0f * k1t + 3.0f * k2t + 4.0f * k3t + 5.0f * k4t + 6.0f * k5t + 7.0f * k6t + 8.0f * k7t + 9.0f * k8t + 10.0f * k9t + 11.0f * k10t + 12.0f * k11t);
            theta = theta - theta_fix;
            if (fabsf(theta_fix) < EPS) { converged = 1; break; }
        }
        scale = tanf(theta) / theta_d;
    } else {
        converged = 1;
    }
    int theta_flipped = (theta_d < 0.0f && theta > 0.0f) || (theta_d > 0.0f && theta < 0.0f);
    if (converged && !theta_flipped) { r.x = px * scale; r.y = py * scale; return r; }
    r.ok = 0; return r;
}
static void genpoly_distort(float x, float y, float z, const KP *p, float *ox, float *oy) {
    const float *k = p->k;
    x = x / z; y = y / z;
    if (genpoly_all_zero(k)) { *ox = x; *oy = y; return; }
    float r = sqrtf(x * x + y * y);
    float theta = atanf(r);
    float theta2  = theta * theta;
    float theta3  = theta2 * theta;
    float theta4  = theta2 * theta2;
    float theta5  = theta2 * theta3;
    float theta6  = theta3 * theta3;
    float theta7  = theta3 * theta4;
    float theta8  = theta4 * theta4;
    float theta9  = theta4 * theta5;
    float theta10 = theta5 * theta5;
    float theta11 = theta5 * theta6;
    float theta12 = theta6 * theta6;
    float theta_d = theta * k[0] + theta2 * k[1] + theta3 * k[2] + theta4 * k[3] + theta5 * k[4] + theta6 * k[5]
                  + theta7 * k[6] + theta8 * k[7] + theta9 * k[8] + theta10 * k[9] + theta11 * k[10] + theta12 * k[11];
    float scale = (r == 0.0f) ? 1.0f : theta_d / r;
    *ox = x * scale; *oy = y * scale;
}

/* gopro.rs:9-29 helpers, :33-49 undistort, :53-67 distort */
static inline float gopro_poly_eval(float p, const float *k) {
    return k[0] + p * (k[1] + p * (k[2] + p * (k[3] + p * (k[4] + p * (k[5] + p * k[6])))));
}
static inline float gopro_poly_deriv(float p, const float *k) {
    return k[1] + p * (2.0f * k[2] + p * (3.0f * k[3] + p * (4.0f * k[4] + p * (5.0f * k[5] + p * (6.0f * k[6])))));
}
static float gopro_poly_invert(float theta, const float *k) {
    float p = (theta - k[0]) / k[1];
    for (int i = 0; i < 10; ++i) {
        float d = gopro_poly_deriv(p, k);
        if (fabsf(d) < 1e-12f) break;
        float fix = (gopro_poly_eval(p, k) - theta) / d;
        p -= fix;
        if (fabsf(fix) < 1e-7f) break;
    }
    return p;
}
#define GOPRO_TMAX 1.5533f
static opt2 gopro_undistort(float px, float py, const KP *p) {
    opt2 r = {1, px, py};
    if (p->k[1] == 0.0f) return r;
    float r_norm = sqrtf(px * px + py * py);
    if (r_norm < 1e-9f) return r;
    float pp = r_norm / p->k[1];
    float theta = gopro_poly_eval(pp, p->k);
    float tt = tanf(GOPRO_TMAX);
    float rr = (theta < GOPRO_TMAX) ? tanf(theta) : tt + (theta - GOPRO_TMAX) * (1.0f + tt * tt);
    float scale = rr / r_norm;
    r.x = px * scale; r.y = py * scale; return r;
}
static void gopro_distort(float x, float y, float z, const KP *p, float *ox, float *oy) {
    float p0 = x / z, p1 = y / z;
    if (p->k[1] == 0.0f) { *ox = p0; *oy = p1; return; }
    float r = sqrtf(p0 * p0 + p1 * p1);
    float tt = tanf(GOPRO_TMAX);
    float theta = (r < tt) ? atanf(r) : GOPRO_TMAX + (r - tt) / (1.0f + tt * tt);
    float pv = gopro_poly_invert(theta, p->k);
    float r_norm = p->k[1] * pv;
    float scale = (r < 1e-9f) ? 1.0f : r_norm / r;
    *ox = p0 * scale; *oy = p1 * scale;
}

/* ---- digital lenses: uv in pixels ------------------------------------- */
typedef void (*map2_fn)(float, float, const float *, float *, float *);

/* gopro_superview.rs:10-17 */
static void superview_map(float u, float v, const float *dp, float *ox, float *oy) {
    (void)dp;
    float x2 = u * u, y2 = v * v;
    *ox = u * (1.2100393f + x2 * (-1.2758402f + x2 * 1.7751845f));
    *oy = v * (0.9364505f + (0.4465308f - 0.7683315f * y2) * y2 + (-0.3574087f + 1.1584653f * y2 + 0.3529348f * x2) * x2);
}
/* gopro6_superview.rs:10-15 */
static void superview6_map(float u, float v, const float *dp, float *ox, float *oy) {
    (void)dp;
    u *= 1.0f - 0.48f * fabsf(u);
    u *= 0.943396f * (1.0f + 0.157895f * fabsf(u));
    v *= 0.943396f * (1.0f + 0.060000f * fabsf(v * 2.0f));
    *ox = u; *oy = v;
}
/* gopro_hyperview.rs:10-17 */
static void hyperview_map(float u, float v, const float *dp, float *ox, float *oy) {
    (void)dp;
    float x2 = u * u, y2 = v * v;
    *ox = u * (1.5805143f + x2 * (-8.1668825f + x2 * (74.5198746f + x2 * (-451.5002441f + x2 * (1551.2922363f + x2 * (-2735.5422363f + x2 * 1923.1572266f))))) + y2 * -0.1086027f);
    *oy = v * (1.0238225f + y2 * -0.1025671f + x2 * (-0.2639930f + x2 * 0.2979266f));
}
/* gopro_warp.rs:10-27 */
static void gopro_warp_map(float u, float v, const float *p, float *ox, float *oy) {
    float x = rs_clamp(u, -0.5f, 0.5f);
    float y = rs_clamp(v, -0.5f, 0.5f);
    float x2 = x * x, y2 = y * y;
    float poly_x = p[0] + x2 * (p[1] + x2 * (p[2] + x2 * (p[3] + x2 * (p[4] + x2 * (p[5] + x2 * p[6])))));
    *ox = x * (poly_x + p[7] * y2) + (u - x);
    *oy = y * (p[8] + p[9] * y2 + p[10] * y2 * y2 + x2 * (p[11] + p[12] * y2 + p[13] * x2)) + (v - y);
}

/* common shape of gopro_superview.rs:21-33 / gopro6_superview.rs:19-29 /
 * gopro_hyperview.rs:21-33 / gopro_warp.rs:31-44 */
static opt2 digital_undistort(int model, float u, float v, const KP *p) {
    opt2 r = {1, u, v};
    const float *dp = p->digital_lens_params;
    if (model == GFW_MODEL_DIGITAL_STRETCH) { /* digital_stretch.rs:12-15 */
        r.x = u / dp[0]; r.y = v / dp[1]; return r;
    }
    float w = (float)p->output_width, h = (float)p->output_height;
    u = (u / w) - 0.5f;
    v = (v / h) - 0.5f;
    float mx, my;
    switch (model) {
    case GFW_MODEL_GOPRO_SUPERVIEW:  superview_map(u, v, dp, &mx, &my);  mx = mx / 1.333333333f; break;
    case GFW_MODEL_GOPRO6_SUPERVIEW: superview6_map(u, v, dp, &mx, &my); break;
    case GFW_MODEL_GOPRO_HYPERVIEW:  hyperview_map(u, v, dp, &mx, &my);  mx = mx / 1.555555555f; break;
    case GFW_MODEL_GOPRO_WARP: {
        float factor = (dp[14] != 0.0f) ? dp[14] : 1.0f;
        gopro_warp_map(u, v, dp, &mx, &my); mx = mx / factor; break; }
    default: return r; /* physical models are not digital lenses */
    }
    r.x = (mx + 0.5f) * w; r.y = (my + 0.5f) * h; return r;
}
/* gopro_superview.rs:37-57 / gopro6_superview.rs:33-51 / gopro_hyperview.rs:37-57 / gopro_warp.rs:48-85 */
static void digital_distort(int model, float x, float y, const KP *p, float *ox, float *oy) {
    const float *dp = p->digital_lens_params;
    if (model == GFW_MODEL_DIGITAL_STRETCH) { /* digital_stretch.rs:19-22 */
        *ox = x * dp[0]; *oy = y * dp[1]; return;
    }
    float sw = (float)p->width, sh = (float)p->height;
    x = (x / sw) - 0.5f;
    y = (y / sh) - 0.5f;
    map2_fn fn; float tx, ty;
    switch (model) {
    case GFW_MODEL_GOPRO_SUPERVIEW:  fn = superview_map;  x = x * 1.333333333f; tx = x; ty = y; break;
    case GFW_MODEL_GOPRO6_SUPERVIEW: fn = superview6_map; tx = x; ty = y; break;
    case GFW_MODEL_GOPRO_HYPERVIEW:  fn = hyperview_map;  x = x * 1.555555555f; tx = x; ty = y; break;
    case GFW_MODEL_GOPRO_WARP: {
        float factor = (dp[14] != 0.0f) ? dp[14] : 1.0f;
        fn = gopro_warp_map; tx = x * factor; ty = y; break; }   /* seed pp stays (x, y) */
    default: *ox = (x + 0.5f) * sw; *oy = (y + 0.5f) * sh; return;
    }
    float ppx = x, ppy = y;
    for (int i = 0; i < 12; ++i) {
        float dx, dy;
        fn(ppx, ppy, dp, &dx, &dy);
        float d0 = dx - tx, d1 = dy - ty;
        if (fabsf(d0) < 1e-6f && fabsf(d1) < 1e-6f) break;
        ppx -= d0; ppy -= d1;
    }
    if (model == GFW_MODEL_GOPRO_WARP) {
        float rx, ry;
        fn(ppx, ppy, dp, &rx, &ry);
        if (fabsf(rx - tx) > 0.02f || fabsf(ry - ty) > 0.02f) { *ox = -99999.0f; *oy = -99999.0f; return; }
    }
    *ox = (ppx + 0.5f) * sw; *oy = (ppy + 0.5f) * sh;
}

/* distortion_models/mod.rs:36-45 enum dispatch.  A digital-lens model used as
 * the *physical* model dispatches to the same functions (z is ignored there). */
static void model_distort(int model, float x, float y, float z, const KP *p, float *ox, float *oy) {
    switch (model) {
    case GFW_MODEL_OPENCV_FISHEYE:     fisheye_distort(x, y, z, p, ox, oy); break;
    case GFW_MODEL_OPENCV_STANDARD:    cvstd_distort(x, y, z, p, ox, oy); break;
    case GFW_MODEL_POLY3:              poly3_distort(x, y, z, p, ox, oy); break;
    case GFW_MODEL_POLY5:              poly5_distort(x, y, z, p, ox, oy); break;
    case GFW_MODEL_PTLENS:             ptlens_distort(x, y, z, p, ox, oy); break;
    case GFW_MODEL_INSTA360:           insta360_distort(x, y, z, p, ox, oy); break;
    case GFW_MODEL_SONY:               sony_distort(x, y, z, p, ox, oy); break;
    case GFW_MODEL_GENERIC_POLYNOMIAL: genpoly_distort(x, y, z, p, ox, oy); break;
    case GFW_MODEL_GOPRO:              gopro_distort(x, y, z, p, ox, oy); break;
    default:                           digital_distort(model, x, y, p, ox, oy); break;
    }
}
static opt2 model_undistort(int model, float x, float y, const KP *p) {
    switch (model) {
    case GFW_MODEL_OPENCV_FISHEYE:     return fisheye_undistort(x, y, p);
    case GFW_MODEL_OPENCV_STANDARD:    return cvstd_undistort(x, y, p);
    case GFW_MODEL_POLY3:              return poly3_undistort(x, y, p);
    case GFW_MODEL_POLY5:              return poly5_undistort(x, y, p);
    case GFW_MODEL_PTLENS:             return ptlens_undistort(x, y, p);
    case GFW_MODEL_INSTA360:           return insta360_undistort(x, y, p);
    case GFW_MODEL_SONY:               return sony_undistort(x, y, p);
    case GFW_MODEL_GENERIC_POLYNOMIAL: return genpoly_undistort(x, y, p);
    case GFW_MODEL_GOPRO:              return gopro_undistort(x, y, p);
    default:                           return digital_undistort(model, x, y, p);
    }
}

/* ---- Sony mesh: gyro_source/splines.rs:88-177, sony.rs:557-563 (f64) ----- */
#define MAX_GRID 9
static void cubic_spline_coefficients(const double *mesh, int n, double size, double *a, double *b, double *c, double *d) {
    double alpha[MAX_GRID], mu[MAX_GRID], z[MAX_GRID];
    double h = size / (double)(n - 1);
    double inv_h = 1.0 / h;
    double three_inv_h = 3.0 * inv_h;
    double h_over_3 = h / 3.0;
    double inv_3h = 1.0 / (3.0 * h);
    for (int i = 0; i < n; ++i) a[i] = mesh[i];
    for (int i = 1; i < n - 1; ++i) alpha[i] = three_inv_h * (a[i + 1] - 2.0 * a[i] + a[i - 1]);
    mu[0] = 0.0; z[0] = 0.0;
    for (int i = 1; i < n - 1; ++i) {
        mu[i] = 1.0 / (4.0 - mu[i - 1]);
        z[i] = (alpha[i] * inv_h - z[i - 1]) * mu[i];
    }
    c[n - 1] = 0.0;
    for (int j = n - 2; j >= 0; --j) {
        c[j] = z[j] - mu[j] * c[j + 1];
        b[j] = (a[j + 1] - a[j]) * inv_h - h_over_3 * (c[j + 1] + 2.0 * c[j]);
        d[j] = (c[j + 1] - c[j]) * inv_3h;
    }
}
static double cubic_spline_interpolate(const double *a, const double *b, const double *c, const double *d, int n, double x, double size) {
    if (x <= 0.0) return a[0] + b[0] * x;
    if (x >= size) {
        double h = size / (double)(n - 1);
        double slope = b[n - 2] + 2.0 * c[n - 2] * h + 3.0 * d[n - 2] * h * h;
        return a[n - 1] + slope * (x - size);
    }
    int64_t i = d2usize(((double)n - 1.0) * x / size);
    if (i > n - 2) i = n - 2;
    double dx = x - size * (double)i / (double)(n - 1);
    return a[i] + b[i] * dx + c[i] * dx * dx + d[i] * dx * dx * dx;
}
static double bivariate_interpolate(int n_x, int n_y, double size_x, double size_y, const double *mesh, int mesh_offset, double x, double y) {
    double iv[MAX_GRID] = {0}, a[MAX_GRID] = {0}, b[MAX_GRID] = {0}, c[MAX_GRID] = {0}, d[MAX_GRID] = {0};
    int64_t i = d2usize(((double)n_x - 1.0) * x / size_x);
    if (i > n_x - 2) i = n_x - 2;
    double dx = x - size_x * (double)i / (double)(n_x - 1);
    double dx2 = dx * dx;
    int grid = MAX_GRID;
    int raw_mesh_len = n_x * n_y * 2;
    int block = grid * 4;
    int coeff_base = 9 + raw_mesh_len + (mesh_offset * n_y * block);
    int64_t offs = coeff_base + i;
    for (int j = 0; j < n_y; ++j) {
        int64_t rb = offs + (int64_t)j * block;
        iv[j] = mesh[rb + (grid * 0)] + mesh[rb + (grid * 1)] * dx + mesh[rb + (grid * 2)] * dx2 + mesh[rb + (grid * 3)] * dx2 * dx;
    }
    cubic_spline_coefficients(iv, n_y, size_y, a, b, c, d);
    return cubic_spline_interpolate(a, b, c, d, n_y, y, size_y);
}

/* ------------------------------------------------------------- context -- */
typedef struct {
    const KP *p;
    const float *matrices;     /* [matrix_count][14] */
    int model, digital;        /* digital = GFW_MODEL_NONE when absent */
    float r_limit_sq;
    const double *mesh; size_t mesh_len;
    float out_c[2], out_f[2];
} wctx;

/* cpu_undistort.rs:133-228 */
static opt2 rotate_and_distort(float px, float py, size_t idx, const wctx *c) {
    const KP *p = c->p;
    const float *m = c->matrices + idx * 14;
    opt2 none = {0, 0, 0};
    float _x = (px * m[0]) + (py * m[1]) + m[2] + p->translation3d[0];
    float _y = (px * m[3]) + (py * m[4]) + m[5] + p->translation3d[1];
    float _w = (px * m[6]) + (py * m[7]) + m[8] + p->translation3d[2];
    if (_w > 0.0f) {
        if (c->r_limit_sq > 0.0f && (_x * _x + _y * _y) > c->r_limit_sq * _w) return none;  /* :139 (sic) */
        if (p->light_refraction_coefficient != 1.0f && p->light_refraction_coefficient > 0.0f) {
            if (_w != 0.0f) {
                float r = sqrtf(_x * _x + _y * _y) / _w;
                float sin_theta_d = (r / sqrtf(1.0f + r * r)) * p->light_refraction_coefficient;
                float r_d = sin_theta_d / sqrtf(1.0f - sin_theta_d * sin_theta_d);
                if (r_d != 0.0f) _w *= r / r_d;
            }
        }
        float u, v;
        model_distort(c->model, _x, _y, _w, p, &u, &v);
        u = u * p->f[0]; v = v * p->f[1];
        if (m[9] != 0.0f || m[10] != 0.0f || m[11] != 0.0f || m[12] != 0.0f || m[13] != 0.0f) {
            float ang_rad = m[11];
            float cos_a = cosf(-ang_rad);
            float sin_a = sinf(-ang_rad);
            float nu = cos_a * u - sin_a * v - m[9]  + m[12];
            float nv = sin_a * u + cos_a * v - m[10] + m[13];
            u = nu; v = nv;
        }
        u = u + p->c[0]; v = v + p->c[1];

        const double *md = c->mesh;
        if (c->mesh_len > 0 && md[0] > 10.0) {                            /* :169-185 */
            double ms0 = md[3], ms1 = md[4];
            float or0 = (float)md[5], or1 = (float)md[6];
            float cs0 = (float)md[7], cs1 = (float)md[8];
            if ((p->flags & 128) == 128) v = (float)p->height - v;
            u = map_coord(u, 0.0f, (float)p->width,  or0, or0 + cs0);
            v = map_coord(v, 0.0f, (float)p->height, or1, or1 + cs1);
            int nx = (int)d2usize(md[1]), ny = (int)d2usize(md[2]);
            double nxp = bivariate_interpolate(nx, ny, ms0, ms1, md, 0, (double)u, (double)v);
            double nyp = bivariate_interpolate(nx, ny, ms0, ms1, md, 1, (double)u, (double)v);
            u = map_coord((float)nxp, or0, or0 + cs0, 0.0f, (float)p->width);
            v = map_coord((float)nyp, or1, or1 + cs1, 0.0f, (float)p->height);
            if ((p->flags & 128) == 128) v = (float)p->height - v;
        }
        if (c->mesh_len > 0 && md[0] > 0.0 && md[d2usize(md[0])] > 0.0) { /* :188-214 focal plane distortion */
            size_t o = (size_t)d2usize(md[0]);
            double ms1 = md[4];
            float or0 = (float)md[5], or1 = (float)md[6];
            float cs0 = (float)md[7], cs1 = (float)md[8];
            double stblz_grid = ms1 / 8.0;
            if ((p->flags & 128) == 128) v = (float)p->height - v;
            u = map_coord(u, 0.0f, (float)p->width,  or0, or0 + cs0);
            v = map_coord(v, 0.0f, (float)p->height, or1, or1 + cs1);
            size_t idx2 = (size_t)d2usize(fmin(fmax(floor((double)v / stblz_grid), 0.0), 7.0));
            double delta = (double)v - stblz_grid * (double)idx2;
            u -= (float)(md[o + 4 + idx2 * 2 + 0] * delta);
            v -= (float)(md[o + 4 + idx2 * 2 + 1] * delta);
            for (size_t j = 0; j < idx2; ++j) {
                u -= (float)(md[o + 4 + j * 2 + 0] * stblz_grid);
                v -= (float)(md[o + 4 + j * 2 + 1] * stblz_grid);
            }
            u = map_coord(u, or0, or0 + cs0, 0.0f, (float)p->width);
            v = map_coord(v, or1, or1 + cs1, 0.0f, (float)p->height);
            if ((p->flags & 128) == 128) v = (float)p->height - v;
        }
        if ((p->flags & 2) == 2 && c->digital != GFW_MODEL_NONE) {       /* :216-220 */
            float du, dv;
            model_distort(c->digital, u, v, 1.0f, p, &du, &dv);
            u = du; v = dv;
        }
        if (p->input_horizontal_stretch > 0.001f) u /= p->input_horizontal_stretch;
        if (p->input_vertical_stretch   > 0.001f) v /= p->input_vertical_stretch;
        opt2 r = {1, u, v};
        return r;
    }
    return none;
}

/* cpu_undistort.rs:262-265 */
static void rotate_point(float px, float py, float angle, float ox, float oy, float o2x, float o2y, float *rx, float *ry) {
    *rx = cosf(angle) * (px - ox) - sinf(angle) * (py - oy) + o2x;
    *ry = sinf(angle) * (px - ox) + cosf(angle) * (py - oy) + o2y;
}

/* cpu_undistort.rs:421-517 */
static opt2 undistort_coord(float opx, float opy, const wctx *c) {
    const KP *p = c->p;
    opt2 none = {0, 0, 0};
    opx = map_coord(opx, (float)p->output_rect[0], (float)(p->output_rect[0] + p->output_rect[2]), 0.0f, (float)p->output_width);
    opy = map_coord(opy, (float)p->output_rect[1], (float)(p->output_rect[1] + p->output_rect[3]), 0.0f, (float)p->output_height);
    opx += p->translation2d[0];
    opy += p->translation2d[1];

    if (p->lens_correction_amount < 1.0f) {                               /* :429-460 */
        float nx = opx, ny = opy;
        if ((p->flags & 2) == 2 && c->digital != GFW_MODEL_NONE) {
            float uzx = (nx - c->out_c[0]) * p->fov + c->out_c[0];
            float uzy = (ny - c->out_c[1]) * p->fov + c->out_c[1];
            opt2 pt = model_undistort(c->digital, uzx, uzy, p);
            if (pt.ok) {
                nx = (pt.x - c->out_c[0]) / p->fov + c->out_c[0];
                ny = (pt.y - c->out_c[1]) / p->fov + c->out_c[1];
            }
        }
        nx = (nx - c->out_c[0]) / c->out_f[0];
        ny = (ny - c->out_c[1]) / c->out_f[1];
        opt2 pt = model_undistort(c->model, nx, ny, p);
        if (pt.ok) { nx = pt.x; ny = pt.y; }
        if (p->light_refraction_coefficient != 1.0f && p->light_refraction_coefficient > 0.0f) {
            float r = sqrtf(nx * nx + ny * ny);
            if (r != 0.0f) {
                float sin_theta_d = (r / sqrtf(1.0f + r * r)) / p->light_refraction_coefficient;
                float r_d = sin_theta_d / sqrtf(1.0f - sin_theta_d * sin_theta_d);
                float factor = r_d / r;
                nx *= factor; ny *= factor;
            }
        }
        nx = (nx * c->out_f[0]) + c->out_c[0];
        ny = (ny * c->out_f[1]) + c->out_c[1];
        opx = nx * (1.0f - p->lens_correction_amount) + (opx * p->lens_correction_amount);
        opy = ny * (1.0f - p->lens_correction_amount) + (opy * p->lens_correction_amount);
    }

    /* :465-479 rolling-shutter row */
    int32_t sy;
    if ((p->flags & 16) == 16) { sy = f2i(roundf(opx)); if (sy > p->width)  sy = p->width;  if (sy < 0) sy = 0; }
    else                       { sy = f2i(roundf(opy)); if (sy > p->height) sy = p->height; if (sy < 0) sy = 0; }
    if (p->matrix_count > 1) {
        size_t idx = (size_t)p->matrix_count / 2;
        opt2 pt = rotate_and_distort(opx, opy, idx, c);
        if (pt.ok) {
            if ((p->flags & 16) == 16) { sy = f2i(roundf(pt.x)); if (sy > p->width)  sy = p->width;  if (sy < 0) sy = 0; }
            else                       { sy = f2i(roundf(pt.y)); if (sy > p->height) sy = p->height; if (sy < 0) sy = 0; }
        }
    }
    size_t idx = (size_t)sy;
    if (idx > (size_t)p->matrix_count - 1) idx = (size_t)p->matrix_count - 1;
    opt2 uv = rotate_and_distort(opx, opy, idx, c);
    if (!uv.ok) return none;

    float fs0 = (float)p->width, fs1 = (float)p->height;
    if (p->input_rotation != 0.0f) {                                      /* :485-491 */
        float rotation = p->input_rotation * (3.14159265358979323846f / 180.0f);
        float s0 = fs0, s1 = fs1;
        rotate_point(s0, s1, rotation, 0.0f, 0.0f, 0.0f, 0.0f, &fs0, &fs1);
        fs0 = roundf(fabsf(fs0)); fs1 = roundf(fabsf(fs1));
        rotate_point(uv.x, uv.y, rotation, s0 / 2.0f, s1 / 2.0f, fs0 / 2.0f, fs1 / 2.0f, &uv.x, &uv.y);
    }
    float width_f = (float)p->width, height_f = (float)p->height;
    if (p->background_mode == 1) {                                        /* :495-499 */
        uv.x = fminf(fmaxf(uv.x, 3.0f), width_f  - 3.0f);
        uv.y = fminf(fmaxf(uv.y, 3.0f), height_f - 3.0f);
    } else if (p->background_mode == 2) {                                 /* :500-509 */
        float rx = roundf(uv.x), ry = roundf(uv.y);
        float width3 = width_f - 3.0f, height3 = height_f - 3.0f;
        if (rx > width3)  uv.x = width3  - (rx - width3);
        if (rx < 3.0f)    uv.x = 3.0f + width_f - (width3  + rx);
        if (ry > height3) uv.y = height3 - (ry - height3);
        if (ry < 3.0f)    uv.y = 3.0f + height_f - (height3 + ry);
    }
    if (p->background_mode != 3) {                                        /* :510-515 */
        uv.x = map_coord(uv.x, 0.0f, fs0, (float)p->source_rect[0], (float)(p->source_rect[0] + p->source_rect[2]));
        uv.y = map_coord(uv.y, 0.0f, fs1, (float)p->source_rect[1], (float)(p->source_rect[1] + p->source_rect[3]));
    }
    return uv;
}

/* ------------------------------------------------ PixelType load / store */
static inline __attribute__((always_inline)) v4 px_to_float(int t, const uint8_t *b) {  /* pixel_formats.rs to_float */
    v4 r = {0, 0, 0, 0};
    uint16_t h[4]; float f[4];
    switch (t) {
    case GFW_PIX_LUMA8:  r.x = (float)b[0]; break;
    case GFW_PIX_LUMA16: memcpy(h, b, 2); r.x = (float)h[0]; break;
    case GFW_PIX_RGB8:   r.x = (float)b[0]; r.y = (float)b[1]; r.z = (float)b[2]; break;
    case GFW_PIX_RGBA8: case GFW_PIX_BGRA8:
        r.x = (float)b[0]; r.y = (float)b[1]; r.z = (float)b[2]; r.w = (float)b[3]; break;
    case GFW_PIX_RGB16:  memcpy(h, b, 6); r.x = (float)h[0]; r.y = (float)h[1]; r.z = (float)h[2]; break;
    case GFW_PIX_RGBA16: case GFW_PIX_AYUV16:
        memcpy(h, b, 8); r.x = (float)h[0]; r.y = (float)h[1]; r.z = (float)h[2]; r.w = (float)h[3]; break;
    case GFW_PIX_RGBAF:  memcpy(f, b, 16); r.x = f[0]; r.y = f[1]; r.z = f[2]; r.w = f[3]; break;
    case GFW_PIX_RGBAF16: memcpy(h, b, 8); r.x = f16_to_f32(h[0]); r.y = f16_to_f32(h[1]); r.z = f16_to_f32(h[2]); r.w = f16_to_f32(h[3]); break;
    case GFW_PIX_R32F:   memcpy(f, b, 4); r.x = f[0]; break;
    case GFW_PIX_UV8:    r.x = (float)b[0]; r.y = (float)b[1]; break;
    case GFW_PIX_UV16:   memcpy(h, b, 4); r.x = (float)h[0]; r.y = (float)h[1]; break;
    }
    return r;
}
static inline __attribute__((always_inline)) void px_from_float(int t, v4 v, uint8_t *b) {  /* pixel_formats.rs from_float */
    uint16_t h[4]; float f[4];
    switch (t) {
    case GFW_PIX_LUMA8:  b[0] = f2u8(v.x); break;
    case GFW_PIX_LUMA16: h[0] = f2u16(v.x); memcpy(b, h, 2); break;
    case GFW_PIX_RGB8:   b[0] = f2u8(v.x); b[1] = f2u8(v.y); b[2] = f2u8(v.z); break;
    case GFW_PIX_RGBA8: case GFW_PIX_BGRA8:
        b[0] = f2u8(v.x); b[1] = f2u8(v.y); b[2] = f2u8(v.z); b[3] = f2u8(v.w); break;
    case GFW_PIX_RGB16:  h[0] = f2u16(v.x); h[1] = f2u16(v.y); h[2] = f2u16(v.z); memcpy(b, h, 6); break;
    case GFW_PIX_RGBA16: case GFW_PIX_AYUV16:
        h[0] = f2u16(v.x); h[1] = f2u16(v.y); h[2] = f2u16(v.z); h[3] = f2u16(v.w); memcpy(b, h, 8); break;
    case GFW_PIX_RGBAF:  f[0] = v.x; f[1] = v.y; f[2] = v.z; f[3] = v.w; memcpy(b, f, 16); break;
    case GFW_PIX_RGBAF16: h[0] = f32_to_f16(v.x); h[1] = f32_to_f16(v.y); h[2] = f32_to_f16(v.z); h[3] = f32_to_f16(v.w); memcpy(b, h, 8); break;
    case GFW_PIX_R32F:   f[0] = v.x; memcpy(b, f, 4); break;
    case GFW_PIX_UV8:    b[0] = f2u8(v.x); b[1] = f2u8(v.y); break;
    case GFW_PIX_UV16:   h[0] = f2u16(v.x); h[1] = f2u16(v.y); memcpy(b, h, 4); break;
    }
}
static const int PIX_BPP[GFW_PIX_COUNT]   = {1, 2, 3, 4, 4, 6, 8, 8, 16, 8, 4, 2, 4};
static const int PIX_COUNT[GFW_PIX_COUNT] = {1, 1, 3, 4, 4, 3, 4, 4, 4, 4, 1, 2, 2};
static const float PIX_MAX[GFW_PIX_COUNT] = {255, 65535, 255, 255, 255, 65535, 65535, 65535, 0, 0, 0, 255, 65535};

static inline v4 v4_scale(v4 a, float s) { v4 r = {a.x * s, a.y * s, a.z * s, a.w * s}; return r; }
static inline v4 v4_add(v4 a, v4 b) { v4 r = {a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w}; return r; }

/* ----- EWA helpers: cpu_undistort.rs:272-326 ---------------------------- */
static void affine_bbox(const float *jac, float *bx, float *by) {
    *bx = 2.0f * fmaxf(fmaxf(fabsf(jac[0] + jac[1]), fabsf(jac[0] - jac[1])), 1.0f);
    *by = 2.0f * fmaxf(fmaxf(fabsf(jac[2] + jac[3]), fabsf(jac[2] - jac[3])), 1.0f);
}
static void clamped_ellipse(const float *jac, float *abc) {
    float jx = jac[0], jy = jac[1], jz = jac[2], jw = jac[3];
    float f0 = fabsf(jx * jw - jy * jz);
    float f = fmaxf(f0 * f0, 0.1f);
    float a = (jz * jz + jw * jw) / f;
    float b = -2.0f * (jx * jz + jy * jw) / f;
    float c = (jx * jx + jy * jy) / f;
    float vx = c - a, vy = -b;
    float lv = sqrtf(vx * vx + vy * vy);
    float v0 = (lv > 0.01f) ? vx / lv : 1.0f;
    float cc = sqrtf(fmaxf(1.0f + v0, 0.0f) / 2.0f);
    float s  = sqrtf(fmaxf(1.0f - v0, 0.0f) / 2.0f);
    float a0 = a * cc * cc - b * cc * s + c * s * s;
    float c0 = a * s * s + b * cc * s + c * cc * cc;
    float bt1 = b * (cc * cc - s * s);
    float bt2 = 2.0f * (a - c) * cc * s;
    float b0 = bt1 + bt2;
    float b0v2 = bt1 - bt2;
    if (fabsf(b0) > fabsf(b0v2)) { s = -s; b0 = b0v2; }
    a0 = fminf(a0, 1.0f);
    c0 = fminf(c0, 1.0f);
    float sn = -s;
    abc[0] = a0 * cc * cc - b0 * cc * sn + c0 * sn * sn;
    abc[1] = 2.0f * a0 * cc * sn + b0 * cc * cc - b0 * sn * sn - 2.0f * c0 * cc * sn;
    abc[2] = a0 * sn * sn + b0 * cc * sn + c0 * cc * cc;
}
static float bc2(float x, const KP *p) {
    x = fabsf(x);
    float x2 = x * x;
    if (x < 1.0f)      return p->ewa_coeffs_p[0] + p->ewa_coeffs_p[1] * x + p->ewa_coeffs_p[2] * x2 + p->ewa_coeffs_p[3] * x2 * x;
    else if (x < 2.0f) return p->ewa_coeffs_q[0] + p->ewa_coeffs_q[1] * x + p->ewa_coeffs_q[2] * x2 + p->ewa_coeffs_q[3] * x2 * x;
    return 0.0f;
}

/* cpu_undistort.rs:329-419 */
static inline __attribute__((always_inline)) v4 sample_input_at(int I, int t, float uvx, float uvy, const float *jac, const uint8_t *input, size_t in_len, const KP *p, v4 bg, int *oob) {
    v4 sum = {0, 0, 0, 0};
    const int bpp = p->bytes_per_pixel;
    if (I > 8) {
        float tx, ty;
        affine_bbox(jac, &tx, &ty);
        int32_t b0 = f2i(floorf(uvx - tx)), b1 = f2i(ceilf(uvx + tx));
        int32_t b2 = f2i(floorf(uvy - ty)), b3 = f2i(ceilf(uvy + ty));
        float sum_div = 0.0f;
        int64_t src_index = (int64_t)b2 * p->stride;
        float abc[3];
        clamped_ellipse(jac, abc);
        for (int32_t in_y = b2; in_y <= b3; ++in_y) {
            float in_fy = (float)in_y - uvy;
            float in_fy2 = in_fy * abc[1];
            float in_fy3 = in_fy * in_fy * abc[2];
            for (int32_t in_x = b0; in_x <= b1; ++in_x) {
                float in_fx = (float)in_x - uvx;
                float dr = in_fx * in_fx * abc[0] + in_fx * in_fy2 + in_fy3;
                float k = bc2(sqrtf(dr), p);
                if (k == 0.0f) continue;
                v4 pixel;
                if (in_y >= p->source_rect[1] && in_y < p->source_rect[1] + p->source_rect[3] && in_x >= p->source_rect[0] && in_x < p->source_rect[0] + p->source_rect[2]) {
                    int64_t off = src_index + (int64_t)bpp * in_x;
                    if (off < 0 || (size_t)(off + bpp) > in_len) { *oob = 1; return sum; }
                    pixel = px_to_float(t, input + off);
                } else pixel = bg;
                sum = v4_add(sum, v4_scale(pixel, k));
                sum_div += k;
            }
            src_index += p->stride;
        }
        sum.x /= sum_div; sum.y /= sum_div; sum.z /= sum_div; sum.w /= sum_div;
    } else {
        const int shift = (I >> 2) + 1;
        static const float OFFS[3] = {0.0f, 1.0f, 3.0f};
        static const int IND[3] = {0, 64, 64 + 128};
        float offset = OFFS[I >> 2];
        int ind = IND[I >> 2];
        float u = uvx - offset, v = uvy - offset;
        int32_t sx0 = f2i(roundf(u * 32.0f));
        int32_t sy0 = f2i(roundf(v * 32.0f));
        int32_t sx = sx0 >> 5, sy = sy0 >> 5;       /* arithmetic shift, as Rust i32 >> */
        const float *coeffs_x = &GFW_COEFFS[ind + (((uint32_t)sx0 & 31u) << shift)];
        const float *coeffs_y = &GFW_COEFFS[ind + (((uint32_t)sy0 & 31u) << shift)];
        int64_t src_index = (int64_t)sy * p->stride + (int64_t)sx * bpp;
        for (int yp = 0; yp < I; ++yp) {
            if (sy + yp >= p->source_rect[1] && sy + yp < p->source_rect[1] + p->source_rect[3]) {
                v4 xsum = {0, 0, 0, 0};
                for (int xp = 0; xp < I; ++xp) {
                    v4 pixel;
                    if (sx + xp >= p->source_rect[0] && sx + xp < p->source_rect[0] + p->source_rect[2]) {
                        int64_t off = src_index + (int64_t)bpp * xp;
                        if (off < 0 || (size_t)(off + bpp) > in_len) { *oob = 1; return sum; }
                        pixel = px_to_float(t, input + off);
                    } else pixel = bg;
                    xsum = v4_add(xsum, v4_scale(pixel, coeffs_x[xp]));
                }
                sum = v4_add(sum, v4_scale(xsum, coeffs_y[yp]));
            } else {
                sum = v4_add(sum, v4_scale(bg, coeffs_y[yp]));
            }
            src_index += p->stride;
        }
    }
    v4 r = { fminf(sum.x, p->pixel_value_limit), fminf(sum.y, p->pixel_value_limit),
             fminf(sum.z, p->pixel_value_limit), fminf(sum.w, p->pixel_value_limit) };
    return r;
}

/* cpu_undistort.rs:254-260 */
static inline void remap_colorrange(v4 *px, int is_y) {
    float s = is_y ? 0.85882352f : 0.87843137f;
    px->x *= s; px->y *= s; px->z *= s; px->w *= s;
    px->x += 16.0f; px->y += 16.0f;
}

static void wctx_init(wctx *c, const KP *p, int model, int digital, const float *matrices, const double *mesh, size_t mesh_len) {
    c->p = p; c->matrices = matrices; c->model = model; c->digital = digital;
    c->r_limit_sq = p->r_limit * p->r_limit;                              /* :521 */
    c->mesh = mesh; c->mesh_len = mesh_len;
    float factor = fmaxf(1.0f - p->lens_correction_amount, 0.001f);       /* :526 */
    c->out_c[0] = (float)p->output_width / 2.0f;
    c->out_c[1] = (float)p->output_height / 2.0f;
    c->out_f[0] = p->f[0] / p->fov / factor;
    c->out_f[1] = p->f[1] / p->fov / factor;
}

/* ------------------------------------------------------------ the row loop */
/* Rows [y0, y1) of one plane: the body of undistort_image_cpu::<I,T>'s par_chunks_mut closure (cpu_undistort.rs:543-626).
 * `I` and `t` reach every use as arguments of an always-inline function: called with literals (WARP_ROWS_INST below) the sampler's tap count and
 * the pixel type's load / store fold at compile time, which is what the Rust original's monomorphisation over <I, T> does; called with run-time
 * values it is the generic form.  Returns the out-of-bounds flag. */
typedef struct {
    const gfw_kernel_params *p; wctx c;
    const uint8_t *input; size_t in_len; uint8_t *output; size_t out_len, ostride;
    int t, I, bpp; int64_t rows;
} plane_job;

static inline __attribute__((always_inline)) int warp_rows(const int I, const int t, const plane_job *J, int64_t y0, int64_t y1) {
    const gfw_kernel_params *p = J->p;
    const wctx *c = &J->c;
    const uint8_t *input = J->input; const size_t in_len = J->in_len;
    const size_t ostride = J->ostride, out_len = J->out_len;
    const int bpp = J->bpp;
    v4 bg = { p->background[0] * p->max_pixel_value, p->background[1] * p->max_pixel_value,
              p->background[2] * p->max_pixel_value, p->background[3] * p->max_pixel_value };     /* :523 */
    const int fill_bg = (p->flags & 4) == 4, fix_range = (p->flags & 1) == 1, is_y = p->plane_index == 0;
    int oob_any = 0;
    for (int64_t y = y0; y < y1; ++y) {
        size_t row_len = ostride; if ((size_t)(y + 1) * ostride > out_len) row_len = out_len - (size_t)y * ostride;
        uint8_t *row = J->output + (size_t)y * ostride;
        const int64_t cols = (int64_t)(row_len / (size_t)bpp);  /* a trailing partial chunk can never be a valid pixel */
        for (int64_t x = 0; x < cols; ++x) {
            float opx = map_coord((float)x, (float)p->output_rect[0], (float)(p->output_rect[0] + p->output_rect[2]), 0.0f, (float)p->output_width);
            float opy = map_coord((float)y, (float)p->output_rect[1], (float)(p->output_rect[1] + p->output_rect[3]), 0.0f, (float)p->output_height);
            if (!(opx >= 0.0f && opy >= 0.0f && f2i(opx) < p->output_width && f2i(opy) < p->output_height)) continue;   /* :551 */
            uint8_t *pix_out = row + (size_t)x * bpp;
            v4 pixel = bg;
            if (fill_bg) { px_from_float(t, bg, pix_out); continue; }
            int oob = 0;
            opt2 uv = undistort_coord((float)x, (float)y, c);
            if (uv.ok) {
                float jac[4] = {1.0f, 0.0f, 0.0f, 1.0f};
                if (I > 8) {                                               /* :567-572 */
                    const float eps = 0.01f;
                    opt2 a = undistort_coord((float)x + eps, (float)y, c);
                    opt2 b = undistort_coord((float)x, (float)y + eps, c);
                    float xyx0 = (a.ok ? a.x : 0.0f) - uv.x, xyx1 = (a.ok ? a.y : 0.0f) - uv.y;
                    float xyy0 = (b.ok ? b.x : 0.0f) - uv.x, xyy1 = (b.ok ? b.y : 0.0f) - uv.y;
                    jac[0] = xyx0 / eps; jac[1] = xyy0 / eps; jac[2] = xyx1 / eps; jac[3] = xyy1 / eps;
                }
                float width_f = (float)p->width, height_f = (float)p->height;
                if (p->background_mode == 3) {                             /* :576-613 */
                    float widthf = width_f - 1.0f, heightf = height_f - 1.0f;
                    float feather = fmaxf(p->background_margin_feather * heightf, 0.0001f);
                    float p2x = uv.x, p2y = uv.y;
                    float alpha = 1.0f;
                    if ((uv.x > widthf - feather) || (uv.x < feather) || (uv.y > heightf - feather) || (uv.y < feather)) {
                        alpha = fmaxf(fminf(fminf(fminf(fminf(widthf - uv.x, heightf - uv.y), uv.x), uv.y) / feather, 1.0f), 0.0f);
                        p2x = p2x / width_f; p2y = p2y / height_f;
                        p2x = ((p2x - 0.5f) * (1.0f - p->background_margin)) + 0.5f;
                        p2y = ((p2y - 0.5f) * (1.0f - p->background_margin)) + 0.5f;
                        p2x = p2x * width_f; p2y = p2y * height_f;
                    }
                    float fs0 = (float)p->width, fs1 = (float)p->height;
                    if (p->input_rotation != 0.0f) {
                        float rotation = p->input_rotation * (3.14159265358979323846f / 180.0f);
                        float s0 = fs0, s1 = fs1;
                        rotate_point(s0, s1, rotation, 0.0f, 0.0f, 0.0f, 0.0f, &fs0, &fs1);
                        fs0 = roundf(fabsf(fs0)); fs1 = roundf(fabsf(fs1));
                    }
                    float sr0 = (float)p->source_rect[0], sr1 = (float)p->source_rect[1];
                    float sr02 = (float)(p->source_rect[0] + p->source_rect[2]), sr13 = (float)(p->source_rect[1] + p->source_rect[3]);
                    float ux = map_coord(uv.x, 0.0f, fs0, sr0, sr02), uy = map_coord(uv.y, 0.0f, fs1, sr1, sr13);
                    p2x = map_coord(p2x, 0.0f, fs0, sr0, sr02); p2y = map_coord(p2y, 0.0f, fs1, sr1, sr13);
                    v4 c1 = sample_input_at(I, t, ux, uy, jac, input, in_len, p, bg, &oob);
                    v4 c2 = sample_input_at(I, t, p2x, p2y, jac, input, in_len, p, bg, &oob);
                    pixel = v4_add(v4_scale(c1, alpha), v4_scale(c2, 1.0f - alpha));
                    if (fix_range) remap_colorrange(&pixel, is_y);
                    px_from_float(t, pixel, pix_out);
                    oob_any |= oob;
                    continue;
                }
                pixel = sample_input_at(I, t, uv.x, uv.y, jac, input, in_len, p, bg, &oob);
            }
            if (fix_range) remap_colorrange(&pixel, is_y);
            px_from_float(t, pixel, pix_out);
            oob_any |= oob;
        }
    }
    return oob_any;
}
/* the instantiations a render meets: the three LUT samplers x the pixel types of the render loop's plane table (rendering/mod.rs:565-649);
 * everything else (EWA, three-channel pixels, RGBAf16 ...) takes the generic form */
typedef int (*warp_rows_fn)(const plane_job *, int64_t, int64_t);
#define WARP_ROWS_INST(I_, T_) static int warp_rows_##I_##_##T_(const plane_job *J, int64_t y0, int64_t y1) { return warp_rows(I_, T_, J, y0, y1); }
#define WARP_ROWS_TYPES(I_) WARP_ROWS_INST(I_, GFW_PIX_LUMA8) WARP_ROWS_INST(I_, GFW_PIX_LUMA16) WARP_ROWS_INST(I_, GFW_PIX_UV8) WARP_ROWS_INST(I_, GFW_PIX_UV16) \
    WARP_ROWS_INST(I_, GFW_PIX_RGBA8) WARP_ROWS_INST(I_, GFW_PIX_RGBA16) WARP_ROWS_INST(I_, GFW_PIX_RGBAF) WARP_ROWS_INST(I_, GFW_PIX_R32F)
WARP_ROWS_TYPES(2) WARP_ROWS_TYPES(4) WARP_ROWS_TYPES(8)
static int warp_rows_generic(const plane_job *J, int64_t y0, int64_t y1) { return warp_rows(J->I, J->t, J, y0, y1); }
static warp_rows_fn warp_rows_pick(int I, int t) {
    if (getenv("GFW_ORACLE_GENERIC")) return warp_rows_generic;          /* tests: the generic form and the instantiations must agree */
#define WARP_ROWS_CASE(I_, T_) if (I == I_ && t == T_) return warp_rows_##I_##_##T_;
#define WARP_ROWS_CASES(I_) WARP_ROWS_CASE(I_, GFW_PIX_LUMA8) WARP_ROWS_CASE(I_, GFW_PIX_LUMA16) WARP_ROWS_CASE(I_, GFW_PIX_UV8) WARP_ROWS_CASE(I_, GFW_PIX_UV16) \
    WARP_ROWS_CASE(I_, GFW_PIX_RGBA8) WARP_ROWS_CASE(I_, GFW_PIX_RGBA16) WARP_ROWS_CASE(I_, GFW_PIX_RGBAF) WARP_ROWS_CASE(I_, GFW_PIX_R32F)
    WARP_ROWS_CASES(2) WARP_ROWS_CASES(4) WARP_ROWS_CASES(8)
    return warp_rows_generic;
}
/* validation of one plane's call + its job; 1 ok, 0 / -1 as gfw_oracle_undistort_image documents */
static int plane_job_init(plane_job *J, const gfw_buffers *buffers, const gfw_kernel_params *p, int pixel_type, int model, int digital, const float *matrices,
                          const double *mesh, size_t mesh_len) {
    if (!buffers || !p || pixel_type < 0 || pixel_type >= GFW_PIX_COUNT) return 0;
    if (buffers->input.kind != GFW_BUF_HOST || buffers->output.kind != GFW_BUF_HOST) return 0;
    if (buffers->output.stride <= 0) return 0;                            /* :534-537 */
    if (p->bytes_per_pixel != PIX_BPP[pixel_type]) return -1;             /* assert_eq! :541 */
    J->p = p; J->input = (const uint8_t *)buffers->input.data; J->output = (uint8_t *)buffers->output.data;
    J->in_len = buffers->input.len; J->out_len = buffers->output.len;
    J->I = p->interpolation; J->t = pixel_type; J->bpp = p->bytes_per_pixel;
    J->ostride = (size_t)buffers->output.stride;                          /* par_chunks_mut(buffers.output.size.2) */
    J->rows = (int64_t)((J->out_len + J->ostride - 1) / J->ostride);
    wctx_init(&J->c, p, model, digital, matrices, mesh, mesh_len);
    return 1;
}

/* ------------------------------------------------------------ public API */
/* undistort_image_cpu::<I,T>: cpu_undistort.rs:233-633.
 * I = params->interpolation (set from Stabilization.interpolation at mod.rs:266,706-714).
 * Returns 1 (true) on success, 0 when a buffer is missing (the `false` arms
 * at :627-632), -1 if the reference would have panicked on an out-of-range
 * slice index. nthreads <= 0: all cores (rayon par_chunks_mut, :543). */
int gfw_oracle_undistort_image(const gfw_buffers *buffers, const gfw_kernel_params *p, int pixel_type,
                               int distortion_model, int digital_lens,
                               const float *matrices, const float *mesh_f32, size_t mesh_len, int nthreads)
{
    double *mesh = NULL;
    if (mesh_len) { mesh = (double *)malloc(mesh_len * sizeof(double)); for (size_t i = 0; i < mesh_len; ++i) mesh[i] = (double)mesh_f32[i]; }  /* :539 */
    plane_job J;
    const int st = plane_job_init(&J, buffers, p, pixel_type, distortion_model, digital_lens, matrices, mesh, mesh_len);
    if (st != 1) { free(mesh); return st; }
    const warp_rows_fn fn = warp_rows_pick(J.I, J.t);
    int oob_any = 0;
#ifdef _OPENMP
    if (nthreads > 0) omp_set_num_threads(nthreads);
#else
    (void)nthreads;
#endif
    #pragma omp parallel for schedule(dynamic, 4) reduction(|:oob_any)
    for (int64_t y = 0; y < J.rows; ++y) oob_any |= fn(&J, y, y + 1);
    free(mesh);
    return oob_any ? -1 : 1;
}

/* The planes of one frame as the CPU BASELINE runs them (bench.py's cpu_baseline leg): the render loop issues the planes one process_pixels call after the
 * other, each a rayon par_chunks_mut over its rows (work stealing over all cores); here the rows of all planes form ONE parallel region, chunks of
 * `chunk` rows dealt round-robin (static: no shared counter for hundreds of threads to fight over, and cheap and dear regions of the frame interleave
 * across the threads).  Same arithmetic, same per-plane results as gfw_oracle_undistort_image — the tests hold both to the reference's fixture. */
int gfw_oracle_undistort_frame(int nplanes, const gfw_buffers *buffers, const gfw_kernel_params *params, const int *pixel_types,
                               int distortion_model, int digital_lens, const float *matrices, int nthreads, int chunk)
{
    if (nplanes < 1 || nplanes > 8 || !buffers || !params || !pixel_types) return 0;
    plane_job J[8]; warp_rows_fn fn[8]; int64_t first_chunk[9];
    if (chunk < 1) chunk = 4;
    first_chunk[0] = 0;
    for (int i = 0; i < nplanes; ++i) {
        const int st = plane_job_init(&J[i], &buffers[i], &params[i], pixel_types[i], distortion_model, digital_lens, matrices, NULL, 0);
        if (st != 1) return st;
        fn[i] = warp_rows_pick(J[i].I, J[i].t);
        first_chunk[i + 1] = first_chunk[i] + (J[i].rows + chunk - 1) / chunk;
    }
    int oob_any = 0;
#ifdef _OPENMP
    if (nthreads > 0) omp_set_num_threads(nthreads);
#else
    (void)nthreads;
#endif
    #pragma omp parallel for schedule(static, 1) reduction(|:oob_any)
    for (int64_t k = 0; k < first_chunk[nplanes]; ++k) {
        int i = 0;
        while (k >= first_chunk[i + 1]) ++i;
        const int64_t y0 = (k - first_chunk[i]) * chunk;
        const int64_t y1 = y0 + chunk < J[i].rows ? y0 + chunk : J[i].rows;
        oob_any |= fn[i](&J[i], y0, y1);
    }
    return oob_any ? -1 : 1;
}

/* Debug taps used by the parity tests to localise a mismatch: the source
 * coordinate (after the source_rect map) the reference computes for output
 * buffer pixel (x,y).  out[0]=ok, out[1]=u, out[2]=v. */
void gfw_oracle_undistort_coord(const gfw_kernel_params *p, int distortion_model, int digital_lens,
                                const float *matrices, const float *mesh_f32, size_t mesh_len,
                                float x, float y, float *out)
{
    double *mesh = NULL;
    if (mesh_len) { mesh = (double *)malloc(mesh_len * sizeof(double)); for (size_t i = 0; i < mesh_len; ++i) mesh[i] = (double)mesh_f32[i]; }
    wctx c; wctx_init(&c, p, distortion_model, digital_lens, matrices, mesh, mesh_len);
    opt2 r = undistort_coord(x, y, &c);
    out[0] = (float)r.ok; out[1] = r.x; out[2] = r.y;
    free(mesh);
}

/* STMap "undist" coordinate map: src/core/stmap.rs:87-109 (the per-pixel closure) + parallel_exr :127-137.
 * coords[(y*width + x)*2 + {0,1}] = rotate_and_distort((x, y), row) when it is Some, else left untouched.
 * `p` is the KernelParams stmap.rs builds (width/height = output size = map size, flags = digital-lens | horizontal-RS). */
int gfw_oracle_stmap_undistort(const gfw_kernel_params *p, int distortion_model, int digital_lens,
                               const float *matrices, const float *mesh_f32, size_t mesh_len,
                               int width, int height, float *coords, int nthreads)
{
    double *mesh = NULL;
    if (mesh_len) { mesh = (double *)malloc(mesh_len * sizeof(double)); for (size_t i = 0; i < mesh_len; ++i) mesh[i] = (double)mesh_f32[i]; }
    wctx c; wctx_init(&c, p, distortion_model, digital_lens, matrices, mesh, mesh_len);
    const int hrs = (p->flags & 16) == 16;
#ifdef _OPENMP
    if (nthreads > 0) omp_set_num_threads(nthreads);
#else
    (void)nthreads;
#endif
    #pragma omp parallel for schedule(dynamic, 4)
    for (int y = 0; y < height; ++y) {
        for (int x = 0; x < width; ++x) {
            const float fx = (float)x, fy = (float)y;
            int32_t sy;
            if (hrs) { sy = f2i(roundf(fx)); if (sy > p->width)  sy = p->width;  if (sy < 0) sy = 0; }
            else     { sy = f2i(roundf(fy)); if (sy > p->height) sy = p->height; if (sy < 0) sy = 0; }
            if (p->matrix_count > 1) {
                opt2 pt = rotate_and_distort(fx, fy, (size_t)p->matrix_count / 2, &c);
                if (pt.ok) {
                    if (hrs) { sy = f2i(roundf(pt.x)); if (sy > p->width)  sy = p->width;  if (sy < 0) sy = 0; }
                    else     { sy = f2i(roundf(pt.y)); if (sy > p->height) sy = p->height; if (sy < 0) sy = 0; }
                }
            }
            size_t idx = (size_t)sy;
            if (idx > (size_t)p->matrix_count - 1) idx = (size_t)p->matrix_count - 1;
            opt2 uv = rotate_and_distort(fx, fy, idx, &c);
            if (uv.ok) { coords[((size_t)y * width + x) * 2 + 0] = uv.x; coords[((size_t)y * width + x) * 2 + 1] = uv.y; }
        }
    }
    free(mesh);
    return 1;
}

/* `r_of` of undistort_points' lens-correction branch (cpu_undistort.rs:804-826): the render's forward map R(o) of an
 * in-frame output pixel o = digital_undistort -> /out_f -> radial undistort -> refraction -> *out_f. */
typedef struct { float out_c[2], out_f[2], amount, factor, fov; int model, digital; const KP *p; } lc_ctx;
static void lc_r_of(const lc_ctx *L, float o0, float o1, float *r0, float *r1) {
    float q0 = o0, q1 = o1;
    if (L->digital != GFW_MODEL_NONE) {
        float uz0 = (q0 - L->out_c[0]) * L->fov + L->out_c[0], uz1 = (q1 - L->out_c[1]) * L->fov + L->out_c[1];
        opt2 d = digital_undistort(L->digital, uz0, uz1, L->p);
        if (d.ok) { q0 = (d.x - L->out_c[0]) / L->fov + L->out_c[0]; q1 = (d.y - L->out_c[1]) / L->fov + L->out_c[1]; }
    }
    float n0 = (q0 - L->out_c[0]) / L->out_f[0], n1 = (q1 - L->out_c[1]) / L->out_f[1];
    opt2 d = model_undistort(L->model, n0, n1, L->p);
    if (d.ok) { n0 = d.x; n1 = d.y; }
    if (L->p->light_refraction_coefficient != 1.0f && L->p->light_refraction_coefficient > 0.0f) {
        float r = sqrtf(n0 * n0 + n1 * n1);
        if (r != 0.0f) {
            float sin_theta_d = (r / sqrtf(1.0f + r * r)) / L->p->light_refraction_coefficient;
            float r_d = sin_theta_d / sqrtf(1.0f - sin_theta_d * sin_theta_d);
            float sc = r_d / r;
            n0 = n0 * sc; n1 = n1 * sc;
        }
    }
    *r0 = (n0 * L->out_f[0]) + L->out_c[0]; *r1 = (n1 * L->out_f[1]) + L->out_c[1];
}
/* cpu_undistort.rs:792-851: solve amount*o + (1-amount)*R(o) = pt for the output position o (Newton, forward differences) */
static void lc_solve(const lc_ctx *L, float *pt0, float *pt1) {
    const float p0 = *pt0, p1 = *pt1;
    float inv0, inv1;
    {
        float n0 = (p0 - L->out_c[0]) / L->out_f[0], n1 = (p1 - L->out_c[1]) / L->out_f[1];
        float d0, d1;
        model_distort(L->model, n0, n1, 1.0f, L->p, &d0, &d1);
        inv0 = (d0 * L->out_f[0]) + L->out_c[0]; inv1 = (d1 * L->out_f[1]) + L->out_c[1];
        if (L->digital != GFW_MODEL_NONE) {
            float uz0 = (inv0 - L->out_c[0]) * L->fov + L->out_c[0], uz1 = (inv1 - L->out_c[1]) * L->fov + L->out_c[1];
            float dd0, dd1;
            digital_distort(L->digital, uz0, uz1, L->p, &dd0, &dd1);
            inv0 = (dd0 - L->out_c[0]) / L->fov + L->out_c[0]; inv1 = (dd1 - L->out_c[1]) / L->fov + L->out_c[1];
        }
    }
    float o0, o1;
    if (isfinite(inv0) && isfinite(inv1)) { o0 = inv0 * L->factor + p0 * L->amount; o1 = inv1 * L->factor + p1 * L->amount; }
    else { o0 = p0; o1 = p1; }
    for (int it = 0; it < 10; ++it) {
        float r0, r1;
        lc_r_of(L, o0, o1, &r0, &r1);
        float g0 = L->amount * o0 + L->factor * r0 - p0, g1 = L->amount * o1 + L->factor * r1 - p1;
        if (fabsf(g0) < 0.02f && fabsf(g1) < 0.02f) break;
        const float eps = 1.0f;
        float rx0, rx1, ry0, ry1;
        lc_r_of(L, o0 + eps, o1, &rx0, &rx1);
        lc_r_of(L, o0, o1 + eps, &ry0, &ry1);
        float j11 = L->amount + L->factor * (rx0 - r0) / eps, j21 = L->factor * (rx1 - r1) / eps;
        float j12 = L->factor * (ry0 - r0) / eps,             j22 = L->amount + L->factor * (ry1 - r1) / eps;
        float det = j11 * j22 - j12 * j21;
        if (!isfinite(det) || fabsf(det) < 1e-9f) break;
        float dx = ( j22 * g0 - j12 * g1) / det;
        float dy = (-j21 * g0 + j11 * g1) / det;
        if (!isfinite(dx) || !isfinite(dy)) break;
        o0 = o0 - dx; o1 = o1 - dy;
    }
    *pt0 = o0; *pt1 = o1;
}

/* Inverse point map: `undistort_points` cpu_undistort.rs:652-858 (the STMap "dist" pass stmap.rs:123-127 and
 * the optical-flow caller :643-650 run it with lens_correction_amount == 1; the zoom search reaches the < 1 branch,
 * :785-851: Newton inverse of the render's blend, taken when p->lens_correction_amount < 1, with p->fov as `fov`).
 * Source-image point -> stabilised output coordinate.
 *   p         the KernelParams `undistort_points` builds (:669-681: width/height/output_*, f, c, k, digital_lens_params,
 *             light_refraction_coefficient) plus input_*_stretch carrying params.lens.input_*_stretch (:704-705)
 *   points    n x 2 f32, or NULL = the pixel grid (x = i % grid_w, y = i / grid_w) that parallel_exr walks
 *   rotations [rotation_count][9] row-major f32: nalgebra::convert::<Matrix3<f64>, Matrix3<f32>> of `new_k * R`
 *             (frame_transform.rs:391-410), index chosen by index_mode: 0 single, 1 point index, 2 grid row, 3 grid column
 *   shifts    optional [rotation_count][5] (sx, sy, angle, ox, oy) (frame_transform.rs:412-440), same indexing
 *   mesh      optional f64 undistorting mesh (file_metadata.mesh_correction[frame].0, passed as f64: :707)
 * The 3x3 * (x, y, 1) product follows nalgebra 0.34's gemv (column axpy: ((r0*x) + r1*y) + r2*1) - third-party
 * arithmetic absent from the tree, restated from its published source; parity unpinned like the rest of the path. */
int gfw_oracle_undistort_points(const gfw_kernel_params *p, int distortion_model, int digital_lens,
                                const float *points, size_t n, int grid_w,
                                const float *rotations, int rotation_count, const float *shifts, int index_mode,
                                const double *md, size_t mesh_len, float *out)
{
    const float c0 = p->c[0], c1 = p->c[1], f0 = p->f[0], f1 = p->f[1];
    /* :683-692 lens-correction blend constants (point-independent) */
    lc_ctx L; memset(&L, 0, sizeof(L));
    const int has_lc = p->lens_correction_amount < 1.0f;
    if (has_lc) {
        L.out_c[0] = (float)p->output_width / 2.0f; L.out_c[1] = (float)p->output_height / 2.0f;
        L.amount = p->lens_correction_amount;
        L.factor = fmaxf(1.0f - L.amount, 0.001f);
        L.out_f[0] = f0 / p->fov / L.factor; L.out_f[1] = f1 / p->fov / L.factor;
        L.fov = p->fov; L.model = distortion_model; L.digital = digital_lens; L.p = p;
    }
    #pragma omp parallel for schedule(static)
    for (size_t i = 0; i < n; ++i) {
        float x, y; size_t gx = 0, gy = 0;
        if (points) { x = points[i * 2]; y = points[i * 2 + 1]; }
        else { gx = i % (size_t)grid_w; gy = i / (size_t)grid_w; x = (float)gx; y = (float)gy; }
        size_t index = index_mode == 1 ? i : index_mode == 2 ? gy : index_mode == 3 ? gx : 0;
        if (index >= (size_t)rotation_count) index = 0;        /* rot_per_point.get(index).unwrap_or(&rr), rr = rotations[0] */
        if (p->input_horizontal_stretch > 0.001f) x *= p->input_horizontal_stretch;        /* :704-705 */
        if (p->input_vertical_stretch   > 0.001f) y *= p->input_vertical_stretch;
        if (digital_lens != GFW_MODEL_NONE) {                                               /* :707-712 */
            opt2 d = digital_undistort(digital_lens, x, y, p);
            if (d.ok) { x = d.x; y = d.y; }
        }
        if (mesh_len > 0) {
            if (md[0] > 0.0 && md[d2usize(md[0])] > 0.0) {                                  /* :715-738 focal plane distortion */
                size_t o = (size_t)d2usize(md[0]);
                double ms1 = md[4];
                float or0 = (float)md[5], or1 = (float)md[6];
                float cs0 = (float)md[7], cs1 = (float)md[8];
                double stblz_grid = ms1 / 8.0;
                x = map_coord(x, 0.0f, (float)p->width,  or0, or0 + cs0);
                y = map_coord(y, 0.0f, (float)p->height, or1, or1 + cs1);
                size_t idx = (size_t)d2usize(fmin(fmax(floor((double)y / stblz_grid), 0.0), 7.0));
                double delta = (double)y - stblz_grid * (double)idx;
                x += (float)(md[o + 4 + idx * 2 + 0] * delta);
                y += (float)(md[o + 4 + idx * 2 + 1] * delta);
                for (size_t j = 0; j < idx; ++j) {
                    x += (float)(md[o + 4 + j * 2 + 0] * stblz_grid);
                    y += (float)(md[o + 4 + j * 2 + 1] * stblz_grid);
                }
                x = map_coord(x, or0, or0 + cs0, 0.0f, (float)p->width);
                y = map_coord(y, or1, or1 + cs1, 0.0f, (float)p->height);
            }
            if (md[0] > 10.0) {                                                             /* :740-752 */
                double ms0 = md[3], ms1 = md[4];
                float or0 = (float)md[5], or1 = (float)md[6];
                float cs0 = (float)md[7], cs1 = (float)md[8];
                x = map_coord(x, 0.0f, (float)p->width,  or0, or0 + cs0);
                y = map_coord(y, 0.0f, (float)p->height, or1, or1 + cs1);
                int nx = (int)d2usize(md[1]), ny = (int)d2usize(md[2]);
                double nxp = bivariate_interpolate(nx, ny, ms0, ms1, md, 0, (double)x, (double)y);
                double nyp = bivariate_interpolate(nx, ny, ms0, ms1, md, 1, (double)x, (double)y);
                x = map_coord((float)nxp, or0, or0 + cs0, 0.0f, (float)p->width);
                y = map_coord((float)nyp, or1, or1 + cs1, 0.0f, (float)p->height);
            }
        }
        if (shifts) {                                                                       /* :754-763 */
            const float *s = shifts + index * 5;
            float cos_a = cosf(s[2]), sin_a = sinf(s[2]);
            x = x - c0 - s[3] + s[0];
            y = y - c1 - s[4] + s[1];
            x = cos_a * x - sin_a * y + c0;
            y = sin_a * x + cos_a * y + c1;                  /* uses the already-rotated x, as the reference does (:761-762) */
        }
        float pwx = (x - c0) / f0, pwy = (y - c1) / f1;                                     /* :765 */
        const float *r = rotations + index * 9;
        opt2 pt = model_undistort(distortion_model, pwx, pwy, p);
        if (pt.ok) {
            float ptx = pt.x, pty = pt.y;
            if (p->light_refraction_coefficient != 1.0f && p->light_refraction_coefficient > 0.0f) {   /* :770-779 */
                float rr = sqrtf(ptx * ptx + pty * pty);
                if (rr != 0.0f) {
                    float sin_theta_d = (rr / sqrtf(1.0f + rr * rr)) / p->light_refraction_coefficient;
                    float r_d = sin_theta_d / sqrtf(1.0f - sin_theta_d * sin_theta_d);
                    float factor = r_d / rr;
                    ptx *= factor; pty *= factor;
                }
            }
            float pr0 = ((r[0] * ptx) + r[1] * pty) + r[2] * 1.0f;                         /* :782 (nalgebra gemv order) */
            float pr1 = ((r[3] * ptx) + r[4] * pty) + r[5] * 1.0f;
            float pr2 = ((r[6] * ptx) + r[7] * pty) + r[8] * 1.0f;
            float q0 = pr0 / pr2, q1 = pr1 / pr2;                                           /* :783 */
            if (has_lc) lc_solve(&L, &q0, &q1);                                              /* :785-851 */
            out[i * 2] = q0; out[i * 2 + 1] = q1;
        } else {
            out[i * 2] = -1000000.0f; out[i * 2 + 1] = -1000000.0f;                         /* :855 */
        }
    }
    return 1;
}

/* Per-model point functions, exposed for the lens-model parity tests. */
void gfw_oracle_distort_point(int model, const gfw_kernel_params *p, float x, float y, float z, float *out) {
    model_distort(model, x, y, z, p, &out[0], &out[1]);
}
int gfw_oracle_undistort_point(int model, const gfw_kernel_params *p, float x, float y, float *out) {
    opt2 r = model_undistort(model, x, y, p);
    out[0] = r.x; out[1] = r.y; return r.ok;
}
/* libm probes: the device math is compared against these on the GPU box. */
void gfw_oracle_libm(int fn, const float *in, float *out, size_t n) {
    for (size_t i = 0; i < n; ++i) {
        switch (fn) {
        case 0: out[i] = atanf(in[i]); break;
        case 1: out[i] = tanf(in[i]); break;
        case 2: out[i] = sinf(in[i]); break;
        case 3: out[i] = cosf(in[i]); break;
        case 4: out[i] = sqrtf(in[i]); break;
        default: out[i] = in[i];
        }
    }
}
int gfw_oracle_pixel_type_info(int t, int *bpp, int *count, float *maxv) {
    if (t < 0 || t >= GFW_PIX_COUNT) return -1;
    *bpp = PIX_BPP[t]; *count = PIX_COUNT[t]; *maxv = PIX_MAX[t]; return 0;
}
int gfw_oracle_num_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
