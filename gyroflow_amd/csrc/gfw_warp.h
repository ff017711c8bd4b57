// gfw_warp.h — device-side per-pixel warp: output pixel -> source coordinate -> taps.
//
// Built for gfx950 with -ffp-contract=off: every f32 operation below is one
// IEEE operation in the order the reference's CPU kernel performs it
// (gyroflow src/core/stabilization/cpu_undistort.rs:133-228, :421-517,
// :329-419 and distortion_models/*.rs), because u8/u16 parity is decided by
// which 1/32-pixel bin round(u*32) falls in.  Uniform parameters arrive in the
// kernel-argument segment (scalar loads -> SGPRs); per-row matrices are read
// from a 64-byte-row table in HBM/L2 (or LDS when staged by the caller).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/gfwarp.h"
#include "gfw_math.h"

#define GFW_EWA_MAX_TAPS (1 << 22)   // EWA bounding-box area (source pixels per output pixel) beyond which bg is written
#define GFW_MAT_STRIDE 16   // floats per matrix row on device: m0..m13, cos(-m11), sin(-m11)

struct GfwPlane {
    gfw_kernel_params p;
    const uint8_t *src;
    uint8_t *dst;
    int64_t dst_len;         // bytes at dst; rows/cols iterate all of it (par_chunks_mut, cpu_undistort.rs:543-544)
    int32_t dst_stride;      // buffers.output.size.2
    int32_t out_rows;        // ceil(dst_len / dst_stride)
    int32_t out_cols;        // dst_stride / bytes_per_pixel
    int32_t pix;             // GFW_PIX_*
    // EWA on planar chroma (round 6): a second plane with the SAME kernel parameters but for its background — U and V of a planar frame — warped by the launch of the
    // first: one set of coordinates, one jacobian, one set of tap weights, two sums (gfw_plane_kernel<.., DUAL>).  nullptr: an ordinary launch.
    const uint8_t *src2;
    uint8_t *dst2;
    float background2[4];    // the second plane's KernelParams::background
};

struct GfwCommon {
    const float *matrices;   // [matrix_count][GFW_MAT_STRIDE]
    const float *mesh;       // f32 mesh data (device) or nullptr
    int32_t mesh_len;
    int32_t model;           // physical lens GFW_MODEL_*
    int32_t digital;         // digital lens or GFW_MODEL_NONE
    int32_t pad_;
    // host-libm-evaluated uniforms (rotate_point's cos/sin of input_rotation, gopro's tan(TMAX))
    float rot_cos, rot_sin;
    float frame_w, frame_h;  // |rotated| frame size (cpu_undistort.rs:488-489), = width/height when rotation == 0
    float gopro_tt;
    float pad2_;
};

// ----------------------------------------------------------------------------
// Rust-semantics scalar helpers
__device__ __forceinline__ int32_t gfw_f2i(float v) {         // `as i32`: trunc, saturate, NaN -> 0
    int32_t r;
    asm("v_cvt_i32_f32 %0, %1" : "=v"(r) : "v"(v));
    return r;
}
__device__ __forceinline__ uint32_t gfw_f2u_sat(float v, float top) {  // `as u8/u16`: trunc, saturate, NaN -> 0
    uint32_t r;
    asm("v_cvt_u32_f32 %0, %1" : "=v"(r) : "v"(v));          // <0 -> 0, NaN -> 0, >2^32-1 -> max
    const uint32_t t = (uint32_t)top;
    return r > t ? t : r;
}
__device__ __forceinline__ float gfw_round(float x) { return roundf(x); }   // half away from zero (ocml)
__device__ __forceinline__ float gfw_min(float a, float b) { return fminf(a, b); }
__device__ __forceinline__ float gfw_max(float a, float b) { return fmaxf(a, b); }
__device__ __forceinline__ float gfw_clampf(float x, float lo, float hi) { if (x < lo) x = lo; if (x > hi) x = hi; return x; }
__device__ __forceinline__ float gfw_map_coord(float x, float in_min, float in_max, float out_min, float out_max) {
    return (x - in_min) * (out_max - out_min) / (in_max - in_min) + out_min;       // util.rs:144-147
}

struct GfwPt { float x, y; bool ok; };

// ----------------------------------------------------------------------------
// Lens models (distortion_models/*.rs).  MODEL is a compile-time id or -1 for
// a wave-uniform run-time switch.
namespace gfw_lens {

__device__ __forceinline__ void fisheye_distort(float x, float y, float z, const float *k, float &ox, float &oy) {
    x = x / z; y = y / z;
    if (k[0] == 0.0f && k[1] == 0.0f && k[2] == 0.0f && k[3] == 0.0f) { ox = x; oy = y; return; }
    const float r = sqrtf(x * x + y * y);
    const float t = gfw_atanf(r);
    const float t2 = t * t, t4 = t2 * t2, t6 = t4 * t2, t8 = t4 * t4;
    const float td = t * (1.0f + k[0] * t2 + k[1] * t4 + k[2] * t6 + k[3] * t8);
    const float s = (r == 0.0f) ? 1.0f : td / r;
    ox = x * s; oy = y * s;
}
__device__ __forceinline__ GfwPt fisheye_undistort(float px, float py, const float *k) {
    GfwPt o{px, py, true};
    if (k[0] == 0.0f && k[1] == 0.0f && k[2] == 0.0f && k[3] == 0.0f) return o;
    const float EPS = 1e-6f, PI = 3.14159265358979323846f;
    float theta_d = sqrtf(px * px + py * py);
    theta_d = gfw_min(gfw_max(theta_d, -PI), PI);
    bool converged = false;
    float theta = theta_d, scale = 0.0f;
    if (gfw_fabsf(theta_d) > EPS) {
        theta = 0.0f;
        for (int i = 0; i < 10; ++i) {
            const float t2 = theta * theta, t4 = t2 * t2, t6 = t4 * t2, t8 = t6 * t2;
            const float a = k[0] * t2, b = k[1] * t4, c = k[2] * t6, d = k[3] * t8;
            float fix = (theta * (1.0f + a + b + c + d) - theta_d) / (1.0f + 3.0f * a + 5.0f * b + 7.0f * c + 9.0f * d);
            fix = gfw_min(gfw_max(fix, -0.9f), 0.9f);
            theta = theta - fix;
            if (gfw_fabsf(fix) < EPS) { converged = true; break; }
        }
        scale = gfw_tanf(theta) / theta_d;
    } else converged = true;
    const bool flipped = (theta_d < 0.0f && theta > 0.0f) || (theta_d > 0.0f && theta < 0.0f);
    if (converged && !flipped) { o.x = px * scale; o.y = py * scale; return o; }
    o.ok = false; return o;
}

__device__ __forceinline__ void cvstd_distort(float x, float y, float z, const float *k, float &ox, float &oy) {
    x = x / z; y = y / z;
    const float r2 = x * x + y * y, r4 = r2 * r2, r6 = r4 * r2;
    const float a1 = 2.0f * x * y, a2 = r2 + 2.0f * x * x, a3 = r2 + 2.0f * y * y;
    const float cdist = 1.0f + k[0] * r2 + k[1] * r4 + k[4] * r6;
    const float icdist2 = 1.0f / (1.0f + k[5] * r2 + k[6] * r4 + k[7] * r6);
    ox = x * cdist * icdist2 + k[2] * a1 + k[3] * a2 + k[8]  * r2 + k[9]  * r4;
    oy = y * cdist * icdist2 + k[2] * a3 + k[3] * a1 + k[10] * r2 + k[11] * r4;
}
__device__ __forceinline__ GfwPt cvstd_undistort(float px, float py, const float *k) {
    GfwPt o{0, 0, true};
    float x = px, y = py;
    for (int i = 0; i < 20; ++i) {
        const float r2 = x * x + y * y;
        const float icdist = (1.0f + ((k[7] * r2 + k[6]) * r2 + k[5]) * r2) / (1.0f + ((k[4] * r2 + k[1]) * r2 + k[0]) * r2);
        if (icdist < 0.0f) { o.ok = false; return o; }
        const float dx = 2.0f * k[2] * x * y + k[3] * (r2 + 2.0f * x * x) + k[8]  * r2 + k[9]  * r2 * r2;
        const float dy = k[2] * (r2 + 2.0f * y * y) + 2.0f * k[3] * x * y + k[10] * r2 + k[11] * r2 * r2;
        x = (px - dx) * icdist;
        y = (py - dy) * icdist;
    }
    o.x = x; o.y = y; return o;
}

#define GFW_NEWTON_EPS 0.00001f
__device__ __forceinline__ void poly3_distort(float x, float y, float z, const float *k, float &ox, float &oy) {
    x = x / z; y = y / z;
    const float poly2 = k[0] * (x * x + y * y) + 1.0f;
    ox = x * poly2; oy = y * poly2;
}
__device__ __forceinline__ GfwPt poly3_undistort(float px, float py, const float *k) {
    GfwPt o{0, 0, false};
    const float inv_k1 = 1.0f / k[0];
    const float rd = sqrtf(px * px + py * py);
    if (rd == 0.0f) return o;
    const float rd_div_k1 = rd * inv_k1;
    float ru = rd;
    for (int i = 0; i < 10; ++i) {
        const float fru = ru * ru * ru + ru * inv_k1 - rd_div_k1;
        if (fru >= -GFW_NEWTON_EPS && fru < GFW_NEWTON_EPS) break;
        if (i > 5) return o;
        ru = ru - (fru / (3.0f * ru * ru + inv_k1));
    }
    if (ru < 0.0f) return o;
    ru = ru / rd;
    o.ok = true; o.x = px * ru; o.y = py * ru; return o;
}
__device__ __forceinline__ void poly5_distort(float x, float y, float z, const float *k, float &ox, float &oy) {
    x = x / z; y = y / z;
    const float ru2 = x * x + y * y;
    const float poly4 = 1.0f + k[0] * ru2 + k[1] * ru2 * ru2;
    ox = x * poly4; oy = y * poly4;
}
__device__ __forceinline__ GfwPt poly5_undistort(float px, float py, const float *k) {
    GfwPt o{0, 0, false};
    const float rd = sqrtf(px * px + py * py);
    if (rd == 0.0f) return o;
    float ru = rd;
    for (int i = 0; i < 10; ++i) {
        const float ru2 = ru * ru;
        const float fru = ru * (1.0f + k[0] * ru2 + k[1] * ru2 * ru2) - rd;
        if (fru >= -GFW_NEWTON_EPS && fru < GFW_NEWTON_EPS) break;
        if (i > 5) return o;
        ru = ru - (fru / (1.0f + 3.0f * k[0] * ru2 + 5.0f * k[1] * ru2 * ru2));
    }
    if (ru < 0.0f) return o;
    ru = ru / rd;
    o.ok = true; o.x = px * ru; o.y = py * ru; return o;
}
__device__ __forceinline__ void ptlens_distort(float x, float y, float z, const float *k, float &ox, float &oy) {
    x = x / z; y = y / z;
    const float ru2 = x * x + y * y;
    const float r = sqrtf(ru2);
    const float poly3 = k[0] * ru2 * r + k[1] * ru2 + k[2] * r + 1.0f;
    ox = x * poly3; oy = y * poly3;
}
__device__ __forceinline__ GfwPt ptlens_undistort(float px, float py, const float *k) {
    GfwPt o{0, 0, false};
    const float rd = sqrtf(px * px + py * py);
    if (rd == 0.0f) return o;
    float ru = rd;
    for (int i = 0; i < 10; ++i) {
        const float fru = ru * (k[0] * ru * ru * ru + k[1] * ru * ru + k[2] * ru + 1.0f) - rd;
        if (fru >= -GFW_NEWTON_EPS && fru < GFW_NEWTON_EPS) break;
        if (i > 5) return o;
        ru = ru - (fru / (4.0f * k[0] * ru * ru * ru + 3.0f * k[1] * ru * ru + 2.0f * k[2] * ru + 1.0f));
    }
    if (ru < 0.0f) return o;
    ru = ru / rd;
    o.ok = true; o.x = px * ru; o.y = py * ru; return o;
}

__device__ __forceinline__ void insta360_distort(float x, float y, float z, const float *k, float &ox, float &oy) {
    const float k1 = k[0], k2 = k[1], k3 = k[2], p1 = k[3], p2 = k[4], xi = k[5];
    const float len = sqrtf(x * x + y * y + z * z);
    x = (x / len) / ((z / len) + xi);
    y = (y / len) / ((z / len) + xi);
    const float r2 = x * x + y * y, r4 = r2 * r2, r6 = r4 * r2;
    ox = x * (1.0f + k1 * r2 + k2 * r4 + k3 * r6) + 2.0f * p1 * x * y + p2 * (r2 + 2.0f * x * x);
    oy = y * (1.0f + k1 * r2 + k2 * r4 + k3 * r6) + 2.0f * p2 * x * y + p1 * (r2 + 2.0f * y * y);
}
__device__ __forceinline__ GfwPt insta360_undistort(float tx, float ty, const float *k) {
    GfwPt o{0, 0, true};
    float px = tx, py = ty;
    for (int i = 0; i < 200; ++i) {
        float dx, dy;
        insta360_distort(px, py, 1.0f, k, dx, dy);
        const float d0 = dx - tx, d1 = dy - ty;
        if (gfw_fabsf(d0) < 1e-6f && gfw_fabsf(d1) < 1e-6f) break;
        px -= d0; py -= d1;
    }
    o.x = px; o.y = py; return o;
}

__device__ __forceinline__ void sony_distort(float x, float y, float z, const float *k, float &ox, float &oy) {
    x = x / z; y = y / z;
    if (k[0] == 0.0f && k[1] == 0.0f && k[2] == 0.0f && k[3] == 0.0f) { ox = x; oy = y; return; }
    const float r = sqrtf(x * x + y * y);
    const float t = gfw_atanf(r);
    const float t2 = t * t, t3 = t2 * t, t4 = t2 * t2, t5 = t2 * t3, t6 = t3 * t3;
    const float td = t * k[0] + t2 * k[1] + t3 * k[2] + t4 * k[3] + t5 * k[4] + t6 * k[5];
    const float s = (r == 0.0f) ? 1.0f : td / r;
    ox = x * s; oy = y * s;
}
__device__ __forceinline__ GfwPt sony_undistort(float px, float py, const float *k) {
    GfwPt o{px, py, true};
    if (k[0] == 0.0f && k[1] == 0.0f && k[2] == 0.0f && k[3] == 0.0f) return o;
    const float EPS = 1e-6f;
    const float theta_d = sqrtf(px * px + py * py);
    bool converged = false;
    float theta = theta_d, scale = 0.0f;
    if (gfw_fabsf(theta_d) > EPS) {
        theta = 0.0f;
        for (int i = 0; i < 10; ++i) {
            const float t2 = theta * theta, t3 = t2 * theta, t4 = t2 * t2, t5 = t2 * t3;
            const float k0 = k[0], a = k[1] * theta, b = k[2] * t2, c = k[3] * t3, d = k[4] * t4, e = k[5] * t5;
            const float fix = (theta * (k0 + a + b + c + d + e) - theta_d) / (k0 + 2.0f * a + 3.0f * b + 4.0f * c + 5.0f * d + 6.0f * e);
            theta = theta - fix;
            if (gfw_fabsf(fix) < EPS) { converged = true; break; }
        }
        scale = gfw_tanf(theta) / theta_d;
    } else converged = true;
    const bool flipped = (theta_d < 0.0f && theta > 0.0f) || (theta_d > 0.0f && theta < 0.0f);
    if (converged && !flipped) { o.x = px * scale; o.y = py * scale; return o; }
    o.ok = false; return o;
}

__device__ __forceinline__ bool genpoly_zero(const float *k) {
    bool z = true;
    #pragma unroll
    for (int i = 0; i < 12; ++i) z = z && (k[i] == 0.0f);
    return z;
}
__device__ __forceinline__ void genpoly_distort(float x, float y, float z, const float *k, float &ox, float &oy) {
    x = x / z; y = y / z;
    if (genpoly_zero(k)) { ox = x; oy = y; return; }
    const float r = sqrtf(x * x + y * y);
    const float t = gfw_atanf(r);
    const float t2 = t * t, t3 = t2 * t, t4 = t2 * t2, t5 = t2 * t3, t6 = t3 * t3, t7 = t3 * t4, t8 = t4 * t4,
                t9 = t4 * t5, t10 = t5 * t5, t11 = t5 * t6, t12 = t6 * t6;
    const float td = t * k[0] + t2 * k[1] + t3 * k[2] + t4 * k[3] + t5 * k[4] + t6 * k[5] + t7 * k[6] + t8 * k[7]
                   + t9 * k[8] + t10 * k[9] + t11 * k[10] + t12 * k[11];
    const float s = (r == 0.0f) ? 1.0f : td / r;
    ox = x * s; oy = y * s;
}
__device__ __forceinline__ GfwPt genpoly_undistort(float px, float py, const float *k) {
    GfwPt o{px, py, true};
    if (genpoly_zero(k)) return o;
    const float EPS = 1e-6f;
    const float theta_d = sqrtf(px * px + py * py);
    bool converged = false;
    float theta = theta_d, scale = 0.0f;
    if (gfw_fabsf(theta_d) > EPS) {
        theta = 0.0f;
        for (int i = 0; i < 10; ++i) {
            const float t = theta;
            const float t2 = t * t, t3 = t2 * t, t4 = t2 * t2, t5 = t2 * t3, t6 = t3 * t3, t7 = t3 * t4, t8 = t4 * t4,
                        t9 = t4 * t5, t10 = t5 * t5, t11 = t5 * t6;
            const float k0 = k[0], a1 = k[1] * t, a2 = k[2] * t2, a3 = k[3] * t3, a4 = k[4] * t4, a5 = k[5] * t5,
                        a6 = k[6] * t6, a7 = k[7] * t7, a8 = k[8] * t8, a9 = k[9] * t9, a10 = k[10] * t10, a11 = k[11] * t11;
            const float fix = (t * (k0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + a8 + a9 + a10 + a11) - theta_d)
                            / (k0 + 2.0f * a1 + 3.0f * a2 + 4.0f * a3 + 5.0f * a4 + 6.0f * a5 + 7.0f * a6 + 8.0f * a7 + 9.0f * a8 + 10.0f * a9 + 11.0f * a10 + 12.0f * a11);
            theta = theta - fix;
            if (gfw_fabsf(fix) < EPS) { converged = true; break; }
        }
        scale = gfw_tanf(theta) / theta_d;
    } else converged = true;
    const bool flipped = (theta_d < 0.0f && theta > 0.0f) || (theta_d > 0.0f && theta < 0.0f);
    if (converged && !flipped) { o.x = px * scale; o.y = py * scale; return o; }
    o.ok = false; return o;
}

__device__ __forceinline__ float gopro_eval(float p, const float *k) {
    return k[0] + p * (k[1] + p * (k[2] + p * (k[3] + p * (k[4] + p * (k[5] + p * k[6])))));
}
__device__ __forceinline__ float gopro_deriv(float p, const float *k) {
    return k[1] + p * (2.0f * k[2] + p * (3.0f * k[3] + p * (4.0f * k[4] + p * (5.0f * k[5] + p * (6.0f * k[6])))));
}
#define GFW_GOPRO_TMAX 1.5533f
__device__ __forceinline__ void gopro_distort(float x, float y, float z, const float *k, float tt, float &ox, float &oy) {
    const float p0 = x / z, p1 = y / z;
    if (k[1] == 0.0f) { ox = p0; oy = p1; return; }
    const float r = sqrtf(p0 * p0 + p1 * p1);
    const float theta = (r < tt) ? gfw_atanf(r) : GFW_GOPRO_TMAX + (r - tt) / (1.0f + tt * tt);
    float p = (theta - k[0]) / k[1];
    for (int i = 0; i < 10; ++i) {
        const float d = gopro_deriv(p, k);
        if (gfw_fabsf(d) < 1e-12f) break;
        const float fix = (gopro_eval(p, k) - theta) / d;
        p -= fix;
        if (gfw_fabsf(fix) < 1e-7f) break;
    }
    const float r_norm = k[1] * p;
    const float s = (r < 1e-9f) ? 1.0f : r_norm / r;
    ox = p0 * s; oy = p1 * s;
}
__device__ __forceinline__ GfwPt gopro_undistort(float px, float py, const float *k, float tt) {
    GfwPt o{px, py, true};
    if (k[1] == 0.0f) return o;
    const float r_norm = sqrtf(px * px + py * py);
    if (r_norm < 1e-9f) return o;
    const float pp = r_norm / k[1];
    const float theta = gopro_eval(pp, k);
    const float rr = (theta < GFW_GOPRO_TMAX) ? gfw_tanf(theta) : tt + (theta - GFW_GOPRO_TMAX) * (1.0f + tt * tt);
    const float s = rr / r_norm;
    o.x = px * s; o.y = py * s; return o;
}

// ---- digital lenses (pixel-space in/out) -----------------------------------
template <int M> __device__ __forceinline__ void dmap(float u, float v, const float *p, float &ox, float &oy) {
    if (M == GFW_MODEL_GOPRO_SUPERVIEW) {
        const float x2 = u * u, y2 = v * v;
        ox = u * (1.2100393f + x2 * (-1.2758402f + x2 * 1.7751845f));
        oy = v * (0.9364505f + (0.4465308f - 0.7683315f * y2) * y2 + (-0.3574087f + 1.1584653f * y2 + 0.3529348f * x2) * x2);
    } else if (M == GFW_MODEL_GOPRO6_SUPERVIEW) {
        u *= 1.0f - 0.48f * gfw_fabsf(u);
        u *= 0.943396f * (1.0f + 0.157895f * gfw_fabsf(u));
        v *= 0.943396f * (1.0f + 0.060000f * gfw_fabsf(v * 2.0f));
        ox = u; oy = v;
    } else if (M == GFW_MODEL_GOPRO_HYPERVIEW) {
        const float x2 = u * u, y2 = v * v;
        ox = u * (1.5805143f + x2 * (-8.1668825f + x2 * (74.5198746f + x2 * (-451.5002441f + x2 * (1551.2922363f + x2 * (-2735.5422363f + x2 * 1923.1572266f))))) + y2 * -0.1086027f);
        oy = v * (1.0238225f + y2 * -0.1025671f + x2 * (-0.2639930f + x2 * 0.2979266f));
    } else {   // GFW_MODEL_GOPRO_WARP
        const float x = gfw_clampf(u, -0.5f, 0.5f), y = gfw_clampf(v, -0.5f, 0.5f);
        const float x2 = x * x, y2 = y * y;
        const float poly_x = p[0] + x2 * (p[1] + x2 * (p[2] + x2 * (p[3] + x2 * (p[4] + x2 * (p[5] + x2 * p[6])))));
        ox = x * (poly_x + p[7] * y2) + (u - x);
        oy = y * (p[8] + p[9] * y2 + p[10] * y2 * y2 + x2 * (p[11] + p[12] * y2 + p[13] * x2)) + (v - y);
    }
}
template <int M> __device__ __forceinline__ void digital_distort_t(float x, float y, const gfw_kernel_params &P, float &ox, float &oy) {
    const float *dp = P.digital_lens_params;
    const float sw = (float)P.width, sh = (float)P.height;
    x = (x / sw) - 0.5f;
    y = (y / sh) - 0.5f;
    float tx, ty;
    if (M == GFW_MODEL_GOPRO_SUPERVIEW) { x = x * 1.333333333f; tx = x; ty = y; }
    else if (M == GFW_MODEL_GOPRO_HYPERVIEW) { x = x * 1.555555555f; tx = x; ty = y; }
    else if (M == GFW_MODEL_GOPRO_WARP) { const float f = (dp[14] != 0.0f) ? dp[14] : 1.0f; tx = x * f; ty = y; }
    else { tx = x; ty = y; }
    float ppx = x, ppy = y;
    for (int i = 0; i < 12; ++i) {
        float dx, dy;
        dmap<M>(ppx, ppy, dp, dx, dy);
        const float d0 = dx - tx, d1 = dy - ty;
        if (gfw_fabsf(d0) < 1e-6f && gfw_fabsf(d1) < 1e-6f) break;
        ppx -= d0; ppy -= d1;
    }
    if (M == GFW_MODEL_GOPRO_WARP) {
        float rx, ry;
        dmap<M>(ppx, ppy, dp, rx, ry);
        if (gfw_fabsf(rx - tx) > 0.02f || gfw_fabsf(ry - ty) > 0.02f) { ox = -99999.0f; oy = -99999.0f; return; }
    }
    ox = (ppx + 0.5f) * sw; oy = (ppy + 0.5f) * sh;
}
template <int M> __device__ __forceinline__ GfwPt digital_undistort_t(float u, float v, const gfw_kernel_params &P) {
    const float *dp = P.digital_lens_params;
    const float w = (float)P.output_width, h = (float)P.output_height;
    u = (u / w) - 0.5f;
    v = (v / h) - 0.5f;
    float mx, my;
    dmap<M>(u, v, dp, mx, my);
    if (M == GFW_MODEL_GOPRO_SUPERVIEW) mx = mx / 1.333333333f;
    else if (M == GFW_MODEL_GOPRO_HYPERVIEW) mx = mx / 1.555555555f;
    else if (M == GFW_MODEL_GOPRO_WARP) { const float f = (dp[14] != 0.0f) ? dp[14] : 1.0f; mx = mx / f; }
    return GfwPt{(mx + 0.5f) * w, (my + 0.5f) * h, true};
}

__device__ __forceinline__ void digital_distort(int model, float x, float y, const gfw_kernel_params &P, float &ox, float &oy) {
    switch (model) {
    case GFW_MODEL_DIGITAL_STRETCH:  ox = x * P.digital_lens_params[0]; oy = y * P.digital_lens_params[1]; break;
    case GFW_MODEL_GOPRO_SUPERVIEW:  digital_distort_t<GFW_MODEL_GOPRO_SUPERVIEW>(x, y, P, ox, oy); break;
    case GFW_MODEL_GOPRO6_SUPERVIEW: digital_distort_t<GFW_MODEL_GOPRO6_SUPERVIEW>(x, y, P, ox, oy); break;
    case GFW_MODEL_GOPRO_HYPERVIEW:  digital_distort_t<GFW_MODEL_GOPRO_HYPERVIEW>(x, y, P, ox, oy); break;
    case GFW_MODEL_GOPRO_WARP:       digital_distort_t<GFW_MODEL_GOPRO_WARP>(x, y, P, ox, oy); break;
    default: ox = ((x / (float)P.width) - 0.5f + 0.5f) * (float)P.width; oy = ((y / (float)P.height) - 0.5f + 0.5f) * (float)P.height; break;
    }
}
__device__ __forceinline__ GfwPt digital_undistort(int model, float u, float v, const gfw_kernel_params &P) {
    switch (model) {
    case GFW_MODEL_DIGITAL_STRETCH:  return GfwPt{u / P.digital_lens_params[0], v / P.digital_lens_params[1], true};
    case GFW_MODEL_GOPRO_SUPERVIEW:  return digital_undistort_t<GFW_MODEL_GOPRO_SUPERVIEW>(u, v, P);
    case GFW_MODEL_GOPRO6_SUPERVIEW: return digital_undistort_t<GFW_MODEL_GOPRO6_SUPERVIEW>(u, v, P);
    case GFW_MODEL_GOPRO_HYPERVIEW:  return digital_undistort_t<GFW_MODEL_GOPRO_HYPERVIEW>(u, v, P);
    case GFW_MODEL_GOPRO_WARP:       return digital_undistort_t<GFW_MODEL_GOPRO_WARP>(u, v, P);
    default: return GfwPt{u, v, true};
    }
}

// enum dispatch of distortion_models/mod.rs:36-45
template <int MODEL>
__device__ __forceinline__ void distort(int model, float x, float y, float z, const gfw_kernel_params &P, const GfwCommon &C, float &ox, float &oy) {
    const int m = (MODEL >= 0) ? MODEL : model;
    switch (m) {
    case GFW_MODEL_OPENCV_FISHEYE:     fisheye_distort(x, y, z, P.k, ox, oy); break;
    case GFW_MODEL_OPENCV_STANDARD:    cvstd_distort(x, y, z, P.k, ox, oy); break;
    case GFW_MODEL_POLY3:              poly3_distort(x, y, z, P.k, ox, oy); break;
    case GFW_MODEL_POLY5:              poly5_distort(x, y, z, P.k, ox, oy); break;
    case GFW_MODEL_PTLENS:             ptlens_distort(x, y, z, P.k, ox, oy); break;
    case GFW_MODEL_INSTA360:           insta360_distort(x, y, z, P.k, ox, oy); break;
    case GFW_MODEL_SONY:               sony_distort(x, y, z, P.k, ox, oy); break;
    case GFW_MODEL_GENERIC_POLYNOMIAL: genpoly_distort(x, y, z, P.k, ox, oy); break;
    case GFW_MODEL_GOPRO:              gopro_distort(x, y, z, P.k, C.gopro_tt, ox, oy); break;
    default:                           digital_distort(m, x, y, P, ox, oy); break;
    }
}
template <int MODEL>
__device__ __forceinline__ GfwPt undistort(int model, float x, float y, const gfw_kernel_params &P, const GfwCommon &C) {
    const int m = (MODEL >= 0) ? MODEL : model;
    switch (m) {
    case GFW_MODEL_OPENCV_FISHEYE:     return fisheye_undistort(x, y, P.k);
    case GFW_MODEL_OPENCV_STANDARD:    return cvstd_undistort(x, y, P.k);
    case GFW_MODEL_POLY3:              return poly3_undistort(x, y, P.k);
    case GFW_MODEL_POLY5:              return poly5_undistort(x, y, P.k);
    case GFW_MODEL_PTLENS:             return ptlens_undistort(x, y, P.k);
    case GFW_MODEL_INSTA360:           return insta360_undistort(x, y, P.k);
    case GFW_MODEL_SONY:               return sony_undistort(x, y, P.k);
    case GFW_MODEL_GENERIC_POLYNOMIAL: return genpoly_undistort(x, y, P.k);
    case GFW_MODEL_GOPRO:              return gopro_undistort(x, y, P.k, C.gopro_tt);
    default:                           return digital_undistort(m, x, y, P);
    }
}
}  // namespace gfw_lens

// ----------------------------------------------------------------------------
// Sony mesh / focal-plane distortion (f64, gyro_source/splines.rs:88-177)
namespace gfw_mesh {
#define GFW_GRID 9
__device__ __forceinline__ int64_t d2us(double v) { if (!(v > 0.0)) return 0; if (v >= 9.2e18) return INT64_MAX; return (int64_t)v; }
__device__ inline double spline_eval(const double *iv, int n, double size, double x) {
    double a[GFW_GRID], b[GFW_GRID], c[GFW_GRID], d[GFW_GRID], alpha[GFW_GRID], mu[GFW_GRID], z[GFW_GRID];
    const double h = size / (double)(n - 1);
    const double inv_h = 1.0 / h, three_inv_h = 3.0 * inv_h, h_over_3 = h / 3.0, inv_3h = 1.0 / (3.0 * h);
    for (int i = 0; i < n; ++i) a[i] = iv[i];
    for (int i = 1; i < n - 1; ++i) alpha[i] = three_inv_h * (a[i + 1] - 2.0 * a[i] + a[i - 1]);
    mu[0] = 0.0; z[0] = 0.0;
    for (int i = 1; i < n - 1; ++i) { mu[i] = 1.0 / (4.0 - mu[i - 1]); z[i] = (alpha[i] * inv_h - z[i - 1]) * mu[i]; }
    c[n - 1] = 0.0;
    for (int j = n - 2; j >= 0; --j) {
        c[j] = z[j] - mu[j] * c[j + 1];
        b[j] = (a[j + 1] - a[j]) * inv_h - h_over_3 * (c[j + 1] + 2.0 * c[j]);
        d[j] = (c[j + 1] - c[j]) * inv_3h;
    }
    if (x <= 0.0) return a[0] + b[0] * x;
    if (x >= size) {
        const double slope = b[n - 2] + 2.0 * c[n - 2] * h + 3.0 * d[n - 2] * h * h;
        return a[n - 1] + slope * (x - size);
    }
    int64_t i = d2us(((double)n - 1.0) * x / size);
    if (i > n - 2) i = n - 2;
    const double dx = x - size * (double)i / (double)(n - 1);
    return a[i] + b[i] * dx + c[i] * dx * dx + d[i] * dx * dx * dx;
}
template <typename MT>
__device__ inline double bivariate(int n_x, int n_y, double size_x, double size_y, const MT *mesh, int mesh_offset, double x, double y) {
    double iv[GFW_GRID];
    for (int j = 0; j < GFW_GRID; ++j) iv[j] = 0.0;
    int64_t i = d2us(((double)n_x - 1.0) * x / size_x);
    if (i > n_x - 2) i = n_x - 2;
    const double dx = x - size_x * (double)i / (double)(n_x - 1);
    const double dx2 = dx * dx;
    const int grid = GFW_GRID, block = grid * 4;
    const int64_t offs = 9 + n_x * n_y * 2 + (mesh_offset * n_y * block) + i;
    for (int j = 0; j < n_y; ++j) {
        const int64_t rb = offs + (int64_t)j * block;
        iv[j] = (double)mesh[rb] + (double)mesh[rb + grid] * dx + (double)mesh[rb + grid * 2] * dx2 + (double)mesh[rb + grid * 3] * dx2 * dx;
    }
    return spline_eval(iv, n_y, size_y, y);
}
}  // namespace gfw_mesh

// ----------------------------------------------------------------------------
// Sony lens-distortion mesh + focal-plane-distortion terms of rotate_and_distort (cpu_undistort.rs:169-214) on the distorted
// point (u, v), after `+ c` and before the digital lens, for the fused kernel's GFW_MODEL_GENERIC_EXTRA instantiation — the same operations as the
// block inside gfw_rotate_and_distort below (per-plane kernels).
__device__ __forceinline__ void gfw_mesh_apply(float &u, float &v, const gfw_kernel_params &P, const GfwCommon &C) {
    if (C.mesh_len > 0) {
        const float *md32 = C.mesh;
        const double md0 = (double)md32[0];
        if (md0 > 10.0) {                                                              // :169-185
            const double ms0 = (double)md32[3], ms1 = (double)md32[4];
            const float or0 = (float)(double)md32[5], or1 = (float)(double)md32[6];
            const float cs0 = (float)(double)md32[7], cs1 = (float)(double)md32[8];
            if ((P.flags & 128) == 128) v = (float)P.height - v;
            u = gfw_map_coord(u, 0.0f, (float)P.width,  or0, or0 + cs0);
            v = gfw_map_coord(v, 0.0f, (float)P.height, or1, or1 + cs1);
            const int nx = (int)gfw_mesh::d2us((double)md32[1]), ny = (int)gfw_mesh::d2us((double)md32[2]);
            const double nxp = gfw_mesh::bivariate(nx, ny, ms0, ms1, md32, 0, (double)u, (double)v);
            const double nyp = gfw_mesh::bivariate(nx, ny, ms0, ms1, md32, 1, (double)u, (double)v);
            u = gfw_map_coord((float)nxp, or0, or0 + cs0, 0.0f, (float)P.width);
            v = gfw_map_coord((float)nyp, or1, or1 + cs1, 0.0f, (float)P.height);
            if ((P.flags & 128) == 128) v = (float)P.height - v;
        }
        if (md0 > 0.0 && (double)md32[gfw_mesh::d2us(md0)] > 0.0) {                   // :188-214
            const int64_t o = gfw_mesh::d2us(md0);
            const double ms1 = (double)md32[4];
            const float or0 = (float)(double)md32[5], or1 = (float)(double)md32[6];
            const float cs0 = (float)(double)md32[7], cs1 = (float)(double)md32[8];
            const double grid = ms1 / 8.0;
            if ((P.flags & 128) == 128) v = (float)P.height - v;
            u = gfw_map_coord(u, 0.0f, (float)P.width,  or0, or0 + cs0);
            v = gfw_map_coord(v, 0.0f, (float)P.height, or1, or1 + cs1);
            const int64_t idx2 = gfw_mesh::d2us(fmin(fmax(floor((double)v / grid), 0.0), 7.0));
            const double delta = (double)v - grid * (double)idx2;
            u -= (float)((double)md32[o + 4 + idx2 * 2 + 0] * delta);
            v -= (float)((double)md32[o + 4 + idx2 * 2 + 1] * delta);
            for (int64_t j = 0; j < idx2; ++j) {
                u -= (float)((double)md32[o + 4 + j * 2 + 0] * grid);
                v -= (float)((double)md32[o + 4 + j * 2 + 1] * grid);
            }
            u = gfw_map_coord(u, or0, or0 + cs0, 0.0f, (float)P.width);
            v = gfw_map_coord(v, or1, or1 + cs1, 0.0f, (float)P.height);
            if ((P.flags & 128) == 128) v = (float)P.height - v;
        }
    }
}

// rotate_and_distort: cpu_undistort.rs:133-228
template <int MODEL>
__device__ __forceinline__ GfwPt gfw_rotate_and_distort(float px, float py, int idx, const gfw_kernel_params &P, const GfwCommon &C, float r_limit_sq) {
    const float *m = C.matrices + (size_t)idx * GFW_MAT_STRIDE;
    const float4 ma = *reinterpret_cast<const float4 *>(m);
    const float4 mb = *reinterpret_cast<const float4 *>(m + 4);
    const float m8 = m[8];
    GfwPt none{0.0f, 0.0f, false};
    const float X = (px * ma.x) + (py * ma.y) + ma.z + P.translation3d[0];
    const float Y = (px * ma.w) + (py * mb.x) + mb.y + P.translation3d[1];
    float W = (px * mb.z) + (py * mb.w) + m8 + P.translation3d[2];
    if (!(W > 0.0f)) return none;
    if (r_limit_sq > 0.0f && (X * X + Y * Y) > r_limit_sq * W) return none;          // :139 (sic)
    if (P.light_refraction_coefficient != 1.0f && P.light_refraction_coefficient > 0.0f) {
        if (W != 0.0f) {
            const float r = sqrtf(X * X + Y * Y) / W;
            const float sin_theta_d = (r / sqrtf(1.0f + r * r)) * P.light_refraction_coefficient;
            const float r_d = sin_theta_d / sqrtf(1.0f - sin_theta_d * sin_theta_d);
            if (r_d != 0.0f) W *= r / r_d;
        }
    }
    float u, v;
    gfw_lens::distort<MODEL>(C.model, X, Y, W, P, C, u, v);
    u = u * P.f[0]; v = v * P.f[1];
    const float4 mc = *reinterpret_cast<const float4 *>(m + 8);     // m8 m9 m10 m11
    const float4 md = *reinterpret_cast<const float4 *>(m + 12);    // m12 m13 cos sin
    if (mc.y != 0.0f || mc.z != 0.0f || mc.w != 0.0f || md.x != 0.0f || md.y != 0.0f) {   // :157-165 IBIS/OIS
        const float cos_a = md.z, sin_a = md.w;                     // host libm cosf(-m11), sinf(-m11)
        const float nu = cos_a * u - sin_a * v - mc.y + md.x;
        const float nv = sin_a * u + cos_a * v - mc.z + md.y;
        u = nu; v = nv;
    }
    u = u + P.c[0]; v = v + P.c[1];

    if (C.mesh_len > 0) {
        const float *md32 = C.mesh;
        const double md0 = (double)md32[0];
        if (md0 > 10.0) {                                                              // :169-185
            const double ms0 = (double)md32[3], ms1 = (double)md32[4];
            const float or0 = (float)(double)md32[5], or1 = (float)(double)md32[6];
            const float cs0 = (float)(double)md32[7], cs1 = (float)(double)md32[8];
            if ((P.flags & 128) == 128) v = (float)P.height - v;
            u = gfw_map_coord(u, 0.0f, (float)P.width,  or0, or0 + cs0);
            v = gfw_map_coord(v, 0.0f, (float)P.height, or1, or1 + cs1);
            const int nx = (int)gfw_mesh::d2us((double)md32[1]), ny = (int)gfw_mesh::d2us((double)md32[2]);
            const double nxp = gfw_mesh::bivariate(nx, ny, ms0, ms1, md32, 0, (double)u, (double)v);
            const double nyp = gfw_mesh::bivariate(nx, ny, ms0, ms1, md32, 1, (double)u, (double)v);
            u = gfw_map_coord((float)nxp, or0, or0 + cs0, 0.0f, (float)P.width);
            v = gfw_map_coord((float)nyp, or1, or1 + cs1, 0.0f, (float)P.height);
            if ((P.flags & 128) == 128) v = (float)P.height - v;
        }
        if (md0 > 0.0 && (double)md32[gfw_mesh::d2us(md0)] > 0.0) {                   // :188-214
            const int64_t o = gfw_mesh::d2us(md0);
            const double ms1 = (double)md32[4];
            const float or0 = (float)(double)md32[5], or1 = (float)(double)md32[6];
            const float cs0 = (float)(double)md32[7], cs1 = (float)(double)md32[8];
            const double grid = ms1 / 8.0;
            if ((P.flags & 128) == 128) v = (float)P.height - v;
            u = gfw_map_coord(u, 0.0f, (float)P.width,  or0, or0 + cs0);
            v = gfw_map_coord(v, 0.0f, (float)P.height, or1, or1 + cs1);
            const int64_t idx2 = gfw_mesh::d2us(fmin(fmax(floor((double)v / grid), 0.0), 7.0));
            const double delta = (double)v - grid * (double)idx2;
            u -= (float)((double)md32[o + 4 + idx2 * 2 + 0] * delta);
            v -= (float)((double)md32[o + 4 + idx2 * 2 + 1] * delta);
            for (int64_t j = 0; j < idx2; ++j) {
                u -= (float)((double)md32[o + 4 + j * 2 + 0] * grid);
                v -= (float)((double)md32[o + 4 + j * 2 + 1] * grid);
            }
            u = gfw_map_coord(u, or0, or0 + cs0, 0.0f, (float)P.width);
            v = gfw_map_coord(v, or1, or1 + cs1, 0.0f, (float)P.height);
            if ((P.flags & 128) == 128) v = (float)P.height - v;
        }
    }
    if ((P.flags & 2) == 2 && C.digital != GFW_MODEL_NONE) {                            // :216-220
        float du, dv;
        gfw_lens::distort<-1>(C.digital, u, v, 1.0f, P, C, du, dv);
        u = du; v = dv;
    }
    if (P.input_horizontal_stretch > 0.001f) u /= P.input_horizontal_stretch;
    if (P.input_vertical_stretch   > 0.001f) v /= P.input_vertical_stretch;
    return GfwPt{u, v, true};
}

// undistort_coord up to (not including) the final source_rect map: cpu_undistort.rs:421-509.
// (x, y) are OUTPUT-BUFFER pixel indices as floats.  The result is in full-resolution source pixels.
// The lens-correction blend of undistort_coord (cpu_undistort.rs:429-460): the output position moves towards its
// undistorted counterpart by (1 - lens_correction_amount) before any projection.
// (lens, digital: the clip's lens model and digital lens — C.model / C.digital, or literals in a run-time specialised kernel)
template <int MODEL>
__device__ __forceinline__ void gfw_lens_correction_blend(float &opx, float &opy, const gfw_kernel_params &P, const GfwCommon &C, const int lens, const int digital) {
        const float factor = gfw_max(1.0f - P.lens_correction_amount, 0.001f);          // :526
        const float ocx = (float)P.output_width / 2.0f, ocy = (float)P.output_height / 2.0f;
        const float ofx = P.f[0] / P.fov / factor, ofy = P.f[1] / P.fov / factor;
        float nx = opx, ny = opy;
        if ((P.flags & 2) == 2 && digital != GFW_MODEL_NONE) {
            const float uzx = (nx - ocx) * P.fov + ocx, uzy = (ny - ocy) * P.fov + ocy;
            const GfwPt pt = gfw_lens::undistort<-1>(digital, uzx, uzy, P, C);
            if (pt.ok) { nx = (pt.x - ocx) / P.fov + ocx; ny = (pt.y - ocy) / P.fov + ocy; }
        }
        nx = (nx - ocx) / ofx; ny = (ny - ocy) / ofy;
        const GfwPt pt = gfw_lens::undistort<MODEL>(lens, nx, ny, P, C);
        if (pt.ok) { nx = pt.x; ny = pt.y; }
        if (P.light_refraction_coefficient != 1.0f && P.light_refraction_coefficient > 0.0f) {
            const float r = sqrtf(nx * nx + ny * ny);
            if (r != 0.0f) {
                const float sin_theta_d = (r / sqrtf(1.0f + r * r)) / P.light_refraction_coefficient;
                const float r_d = sin_theta_d / sqrtf(1.0f - sin_theta_d * sin_theta_d);
                const float fac = r_d / r;
                nx *= fac; ny *= fac;
            }
        }
        nx = (nx * ofx) + ocx; ny = (ny * ofy) + ocy;
        opx = nx * (1.0f - P.lens_correction_amount) + (opx * P.lens_correction_amount);
        opy = ny * (1.0f - P.lens_correction_amount) + (opy * P.lens_correction_amount);
}
template <int MODEL>
__device__ __forceinline__ void gfw_lens_correction_blend(float &opx, float &opy, const gfw_kernel_params &P, const GfwCommon &C) {
    gfw_lens_correction_blend<MODEL>(opx, opy, P, C, C.model, C.digital);
}
template <int MODEL>
__device__ __forceinline__ GfwPt gfw_undistort_coord_fullres(float x, float y, const gfw_kernel_params &P, const GfwCommon &C) {
    float opx = gfw_map_coord(x, (float)P.output_rect[0], (float)(P.output_rect[0] + P.output_rect[2]), 0.0f, (float)P.output_width);
    float opy = gfw_map_coord(y, (float)P.output_rect[1], (float)(P.output_rect[1] + P.output_rect[3]), 0.0f, (float)P.output_height);
    opx += P.translation2d[0];
    opy += P.translation2d[1];
    const float r_limit_sq = P.r_limit * P.r_limit;

    if (P.lens_correction_amount < 1.0f) gfw_lens_correction_blend<MODEL>(opx, opy, P, C);   // :429-460

    const bool hrs = (P.flags & 16) == 16;                                              // :465-479
    const int32_t lim = hrs ? P.width : P.height;
    int32_t sy = gfw_f2i(gfw_round(hrs ? opx : opy));
    sy = max(min(sy, lim), 0);
    if (P.matrix_count > 1) {
        const GfwPt pt = gfw_rotate_and_distort<MODEL>(opx, opy, P.matrix_count / 2, P, C, r_limit_sq);
        if (pt.ok) { sy = gfw_f2i(gfw_round(hrs ? pt.x : pt.y)); sy = max(min(sy, lim), 0); }
    }
    const int idx = min(sy, P.matrix_count - 1);
    GfwPt uv = gfw_rotate_and_distort<MODEL>(opx, opy, idx, P, C, r_limit_sq);
    if (!uv.ok) return uv;

    if (P.input_rotation != 0.0f) {                                                     // :485-491 rotate_point(uv, rot, size/2, frame_size/2)
        const float ox = (float)P.width / 2.0f, oy = (float)P.height / 2.0f;
        const float o2x = C.frame_w / 2.0f, o2y = C.frame_h / 2.0f;
        const float rx = C.rot_cos * (uv.x - ox) - C.rot_sin * (uv.y - oy) + o2x;
        const float ry = C.rot_sin * (uv.x - ox) + C.rot_cos * (uv.y - oy) + o2y;
        uv.x = rx; uv.y = ry;
    }
    const float width_f = (float)P.width, height_f = (float)P.height;
    if (P.background_mode == 1) {                                                       // :495-499
        uv.x = gfw_min(gfw_max(uv.x, 3.0f), width_f  - 3.0f);
        uv.y = gfw_min(gfw_max(uv.y, 3.0f), height_f - 3.0f);
    } else if (P.background_mode == 2) {                                                // :500-509
        const float rx = gfw_round(uv.x), ry = gfw_round(uv.y);
        const float width3 = width_f - 3.0f, height3 = height_f - 3.0f;
        if (rx > width3)  uv.x = width3  - (rx - width3);
        if (rx < 3.0f)    uv.x = 3.0f + width_f - (width3  + rx);
        if (ry > height3) uv.y = height3 - (ry - height3);
        if (ry < 3.0f)    uv.y = 3.0f + height_f - (height3 + ry);
    }
    return uv;
}
// the final map into the plane's source_rect (:510-515; also :599-602 for mode 3)
__device__ __forceinline__ void gfw_to_source_rect(float &u, float &v, const gfw_kernel_params &P, const GfwCommon &C) {
    u = gfw_map_coord(u, 0.0f, C.frame_w, (float)P.source_rect[0], (float)(P.source_rect[0] + P.source_rect[2]));
    v = gfw_map_coord(v, 0.0f, C.frame_h, (float)P.source_rect[1], (float)(P.source_rect[1] + P.source_rect[3]));
}

// ----------------------------------------------------------------------------
// PixelType load/store (pixel_formats.rs).  N = element count; unused lanes are not computed.
template <int PIX> struct GfwPix;
#define GFW_DEF_PIX(ID, N_, BPP_, SCALAR, TOP)                                                         \
    template <> struct GfwPix<ID> {                                                                    \
        static constexpr int N = N_; static constexpr int BPP = BPP_;                                  \
        __device__ static __forceinline__ void load(const uint8_t *p, float *v) {                      \
            const SCALAR *s = reinterpret_cast<const SCALAR *>(p);                                     \
            _Pragma("unroll") for (int i = 0; i < N_; ++i) v[i] = (float)s[i];                         \
        }                                                                                              \
        __device__ static __forceinline__ void store(uint8_t *p, const float *v) {                     \
            SCALAR *s = reinterpret_cast<SCALAR *>(p);                                                 \
            _Pragma("unroll") for (int i = 0; i < N_; ++i) s[i] = (SCALAR)gfw_f2u_sat(v[i], TOP);      \
        }                                                                                              \
    };
GFW_DEF_PIX(GFW_PIX_LUMA8,  1, 1, uint8_t,  255.0f)
GFW_DEF_PIX(GFW_PIX_LUMA16, 1, 2, uint16_t, 65535.0f)
GFW_DEF_PIX(GFW_PIX_RGB8,   3, 3, uint8_t,  255.0f)
GFW_DEF_PIX(GFW_PIX_RGBA8,  4, 4, uint8_t,  255.0f)
GFW_DEF_PIX(GFW_PIX_BGRA8,  4, 4, uint8_t,  255.0f)
GFW_DEF_PIX(GFW_PIX_RGB16,  3, 6, uint16_t, 65535.0f)
GFW_DEF_PIX(GFW_PIX_RGBA16, 4, 8, uint16_t, 65535.0f)
GFW_DEF_PIX(GFW_PIX_AYUV16, 4, 8, uint16_t, 65535.0f)
GFW_DEF_PIX(GFW_PIX_UV8,    2, 2, uint8_t,  255.0f)
GFW_DEF_PIX(GFW_PIX_UV16,   2, 4, uint16_t, 65535.0f)
template <> struct GfwPix<GFW_PIX_RGBAF> {
    static constexpr int N = 4; static constexpr int BPP = 16;
    __device__ static __forceinline__ void load(const uint8_t *p, float *v) { const float *s = reinterpret_cast<const float *>(p); v[0] = s[0]; v[1] = s[1]; v[2] = s[2]; v[3] = s[3]; }
    __device__ static __forceinline__ void store(uint8_t *p, const float *v) { float *s = reinterpret_cast<float *>(p); s[0] = v[0]; s[1] = v[1]; s[2] = v[2]; s[3] = v[3]; }
};
template <> struct GfwPix<GFW_PIX_R32F> {
    static constexpr int N = 1; static constexpr int BPP = 4;
    __device__ static __forceinline__ void load(const uint8_t *p, float *v) { v[0] = *reinterpret_cast<const float *>(p); }
    __device__ static __forceinline__ void store(uint8_t *p, const float *v) { *reinterpret_cast<float *>(p) = v[0]; }
};
template <> struct GfwPix<GFW_PIX_RGBAF16> {   // half 2.7.1 from_f32/to_f32 = IEEE RNE = v_cvt_f16_f32 / v_cvt_f32_f16
    static constexpr int N = 4; static constexpr int BPP = 8;
    __device__ static __forceinline__ void load(const uint8_t *p, float *v) { const _Float16 *s = reinterpret_cast<const _Float16 *>(p); for (int i = 0; i < 4; ++i) v[i] = (float)s[i]; }
    __device__ static __forceinline__ void store(uint8_t *p, const float *v) { _Float16 *s = reinterpret_cast<_Float16 *>(p); for (int i = 0; i < 4; ++i) s[i] = (_Float16)v[i]; }
};

// ----------------------------------------------------------------------------
// sample_input_at: cpu_undistort.rs:329-419.  I in {2,4,8}: LUT taps; I == 0: EWA.
template <int PIX, int I, bool DUAL = false>
__device__ __forceinline__ void gfw_sample(float uvx, float uvy, const float *jac, const gfw_kernel_params &P, const uint8_t *src,
                                           const float *bg, const float *lut /* LDS or global COEFFS */, float *out,
                                           const uint8_t *src2 = nullptr, const float *bg2 = nullptr, float *out2 = nullptr) {
    constexpr int N = GfwPix<PIX>::N;
    constexpr int BPP = GfwPix<PIX>::BPP;
    static_assert(!DUAL || I == 0, "the second plane rides on the EWA sampler only");
    float sum[N], sum2[N];       // DUAL: the second plane's sums — same taps, same weights, same order of additions as a launch of its own
    #pragma unroll
    for (int c = 0; c < N; ++c) { sum[c] = 0.0f; sum2[c] = 0.0f; }
    const int sr0 = P.source_rect[0], sr1 = P.source_rect[1];
    const int sr0e = P.source_rect[0] + P.source_rect[2], sr1e = P.source_rect[1] + P.source_rect[3];
    if constexpr (I == 0) {
        // affine_bbox / clamped_ellipse / bc2: cpu_undistort.rs:272-326
        const float jx = jac[0], jy = jac[1], jz = jac[2], jw = jac[3];
        const float tx = 2.0f * gfw_max(gfw_max(gfw_fabsf(jx + jy), gfw_fabsf(jx - jy)), 1.0f);
        const float ty = 2.0f * gfw_max(gfw_max(gfw_fabsf(jz + jw), gfw_fabsf(jz - jw)), 1.0f);
        const int b0 = gfw_f2i(floorf(uvx - tx)), b1 = gfw_f2i(ceilf(uvx + tx));
        const int b2 = gfw_f2i(floorf(uvy - ty)), b3 = gfw_f2i(ceilf(uvy + ty));
        const float f0 = gfw_fabsf(jx * jw - jy * jz);
        const float f = gfw_max(f0 * f0, 0.1f);
        const float a = (jz * jz + jw * jw) / f;
        const float b = -2.0f * (jx * jz + jy * jw) / f;
        const float c = (jx * jx + jy * jy) / f;
        const float vx = c - a, vy = -b;
        const float lv = sqrtf(vx * vx + vy * vy);
        const float v0 = (lv > 0.01f) ? vx / lv : 1.0f;
        const float cc = sqrtf(gfw_max(1.0f + v0, 0.0f) / 2.0f);
        float s = sqrtf(gfw_max(1.0f - v0, 0.0f) / 2.0f);
        float a0 = a * cc * cc - b * cc * s + c * s * s;
        float c0 = a * s * s + b * cc * s + c * cc * cc;
        const float bt1 = b * (cc * cc - s * s);
        const float bt2 = 2.0f * (a - c) * cc * s;
        float b0v = bt1 + bt2;
        const float b0v2 = bt1 - bt2;
        if (gfw_fabsf(b0v) > gfw_fabsf(b0v2)) { s = -s; b0v = b0v2; }
        a0 = gfw_min(a0, 1.0f);
        c0 = gfw_min(c0, 1.0f);
        const float sn = -s;
        const float A = a0 * cc * cc - b0v * cc * sn + c0 * sn * sn;
        const float B = 2.0f * a0 * cc * sn + b0v * cc * cc - b0v * sn * sn - 2.0f * c0 * cc * sn;
        const float Cc = a0 * sn * sn + b0v * cc * sn + c0 * cc * cc;
        float sum_div = 0.0f;
        // Guard, not in the reference: its bounding box is unbounded (`as i32` saturates at +-2^31), so a degenerate
        // jacobian makes the CPU loop run for hours and would hang a GPU queue.  Footprints above GFW_EWA_MAX_TAPS source
        // pixels per output pixel are written as background instead.
        const bool too_large = ((int64_t)b1 - b0 + 1) * ((int64_t)b3 - b2 + 1) > (int64_t)GFW_EWA_MAX_TAPS;
        // Both loops run over INCLUSIVE ranges whose ends may be INT_MAX (gfw_f2i saturates: coordinates at or beyond 2^31, an infinite jacobian entry) — `in_x <= b1`
        // would never turn false there and `b2 - 1` would wrap at INT_MIN; the reference's `bounds.0..=bounds.1` visits each index once.  So: leave AT the last index
        // (the counter is advanced only when another index follows; `continue` reaches the advance like any other path).
        for (int in_y = b2, more_y = (!too_large && b2 <= b3) ? 1 : 0; more_y; more_y = in_y != b3, in_y += more_y) {
            const float in_fy = (float)in_y - uvy;
            const float in_fy2 = in_fy * B;
            const float in_fy3 = in_fy * in_fy * Cc;
            const bool yin = in_y >= sr1 && in_y < sr1e;
            for (int in_x = b0, more_x = b0 <= b1 ? 1 : 0; more_x; more_x = in_x != b1, in_x += more_x) {
                const float in_fx = (float)in_x - uvx;
                const float dr = in_fx * in_fx * A + in_fx * in_fy2 + in_fy3;
                // (round 6) a tap outside the filter's support leaves before its root is taken: a correctly rounded root is below 2 exactly when its operand is
                // below 4 (sqrt(4 - 2^-22) = 2 - 2^-24 - 2^-50..., under the midpoint of the last two floats below 2), and a NaN fails both forms of the test — so
                // this is `kk == 0 -> continue` of the reference for those taps, decided 25 instructions earlier.  Neighbouring lanes sit at neighbouring
                // offsets of the same tap, so the corners of the bounding box (55-60 % of it for a near-identity jacobian) are dead for whole waves.
                if (!(dr < 4.0f)) continue;
                float xx = gfw_fabsf(sqrtf(dr));
                const float x2 = xx * xx;
                float kk = 0.0f;
                if (xx < 1.0f)      kk = P.ewa_coeffs_p[0] + P.ewa_coeffs_p[1] * xx + P.ewa_coeffs_p[2] * x2 + P.ewa_coeffs_p[3] * x2 * xx;
                else if (xx < 2.0f) kk = P.ewa_coeffs_q[0] + P.ewa_coeffs_q[1] * xx + P.ewa_coeffs_q[2] * x2 + P.ewa_coeffs_q[3] * x2 * xx;
                if (kk == 0.0f) continue;
                float px[N], qx[N];
                if (yin && in_x >= sr0 && in_x < sr0e) {
                    GfwPix<PIX>::load(src + (int64_t)in_y * P.stride + (int64_t)in_x * BPP, px);
                    if constexpr (DUAL) GfwPix<PIX>::load(src2 + (int64_t)in_y * P.stride + (int64_t)in_x * BPP, qx);
                } else { _Pragma("unroll") for (int c2 = 0; c2 < N; ++c2) { px[c2] = bg[c2]; if constexpr (DUAL) qx[c2] = bg2[c2]; } }
                #pragma unroll
                for (int c2 = 0; c2 < N; ++c2) { sum[c2] = sum[c2] + kk * px[c2]; if constexpr (DUAL) sum2[c2] = sum2[c2] + kk * qx[c2]; }
                sum_div += kk;
            }
        }
        #pragma unroll
        for (int c2 = 0; c2 < N; ++c2) { sum[c2] = too_large ? bg[c2] : sum[c2] / sum_div; if constexpr (DUAL) sum2[c2] = too_large ? bg2[c2] : sum2[c2] / sum_div; }
    } else {
        constexpr int SHIFT = (I >> 2) + 1;
        constexpr float OFFSET = (I == 2) ? 0.0f : (I == 4 ? 1.0f : 3.0f);
        constexpr int IND = (I == 2) ? 0 : (I == 4 ? 64 : 192);
        const float u = uvx - OFFSET, v = uvy - OFFSET;
        const int sx0 = gfw_f2i(gfw_round(u * 32.0f));
        const int sy0 = gfw_f2i(gfw_round(v * 32.0f));
        const int sx = sx0 >> 5, sy = sy0 >> 5;
        constexpr int IT = I > 0 ? I : 1;
        float cx[IT], cy[IT];
        if (I == 2) {
            // phases are k/32 exactly: {1 - k/32, k/32} equals the table row (cpu_undistort.rs:14-19)
            cx[1] = (float)(sx0 & 31) * 0.03125f; cx[0] = 1.0f - cx[1];
            cy[1] = (float)(sy0 & 31) * 0.03125f; cy[0] = 1.0f - cy[1];
        } else {
            const float *tx = lut + IND + (((uint32_t)sx0 & 31u) << SHIFT);
            const float *ty = lut + IND + (((uint32_t)sy0 & 31u) << SHIFT);
            #pragma unroll
            for (int i = 0; i < I; ++i) { cx[i] = tx[i]; cy[i] = ty[i]; }
        }
        const uint8_t *row = src + (int64_t)sy * P.stride + (int64_t)sx * BPP;
        #pragma unroll
        for (int yp = 0; yp < I; ++yp) {
            if (sy + yp >= sr1 && sy + yp < sr1e) {
                float xs[N];
                #pragma unroll
                for (int c = 0; c < N; ++c) xs[c] = 0.0f;
                #pragma unroll
                for (int xp = 0; xp < I; ++xp) {
                    float px[N];
                    if (sx + xp >= sr0 && sx + xp < sr0e) GfwPix<PIX>::load(row + xp * BPP, px);
                    else { _Pragma("unroll") for (int c = 0; c < N; ++c) px[c] = bg[c]; }
                    #pragma unroll
                    for (int c = 0; c < N; ++c) xs[c] = xs[c] + px[c] * cx[xp];
                }
                #pragma unroll
                for (int c = 0; c < N; ++c) sum[c] = sum[c] + xs[c] * cy[yp];
            } else {
                #pragma unroll
                for (int c = 0; c < N; ++c) sum[c] = sum[c] + bg[c] * cy[yp];
            }
            row += P.stride;
        }
    }
    #pragma unroll
    for (int c = 0; c < N; ++c) { out[c] = gfw_min(sum[c], P.pixel_value_limit); if constexpr (DUAL) out2[c] = gfw_min(sum2[c], P.pixel_value_limit); }
}

// remap_colorrange: cpu_undistort.rs:254-260 (only the first N lanes exist)
template <int N> __device__ __forceinline__ void gfw_remap_colorrange(float *px, bool is_y) {
    const float s = is_y ? 0.85882352f : 0.87843137f;
    #pragma unroll
    for (int c = 0; c < N; ++c) px[c] *= s;
    px[0] += 16.0f;
    if (N > 1) px[1] += 16.0f;
}
