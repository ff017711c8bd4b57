// gfw_math.h — fixed-operation-sequence f32 math shared by every gfwarp kernel.
//
// The reference (Rust std on linux-gnu) gets atan/tan from the system libm;
// on this image that is glibc 2.35, whose float atanf/tanf are the fdlibm-
// lineage routines made only of IEEE f32 +,-,*,/ on fixed constants.  Written
// here as the same operation sequence, compiled WITHOUT fp contraction
// (-ffp-contract=off), they are bit-identical to libm on every input — which
// tests/test_math_host.py checks exhaustively on the host build of this very
// header and tests/test_gpu_math.py re-checks on the device build.
//
// Algorithms: the published fdlibm float routines (s_atanf / k_tanf /
// e_rem_pio2f, Sun Microsystems 1993, as shipped in glibc <= 2.40
// sysdeps/ieee754/flt-32).
#pragma once
#include <stdint.h>
#include <string.h>

#if defined(__HIPCC__) || defined(__HIP_DEVICE_COMPILE__)
#define GFW_HD __host__ __device__ __forceinline__
#else
#define GFW_HD static inline
#endif

GFW_HD uint32_t gfw_f2u(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
GFW_HD float gfw_u2f(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }
GFW_HD float gfw_fabsf(float x) { return gfw_u2f(gfw_f2u(x) & 0x7fffffffu); }

// atanf: argument reduction to one of 5 intervals (one IEEE division), then an
// 11-term odd/even split polynomial in z = x*x.
GFW_HD float gfw_atanf(float x) {
    const float hi0 = 4.6364760399e-01f, hi1 = 7.8539812565e-01f, hi2 = 9.8279368877e-01f, hi3 = 1.5707962513e+00f;
    const float lo0 = 5.0121582440e-09f, lo1 = 3.7748947079e-08f, lo2 = 3.4473217170e-08f, lo3 = 7.5497894159e-08f;
    const float aT0 = 3.3333334327e-01f, aT1 = -2.0000000298e-01f, aT2 = 1.4285714924e-01f, aT3 = -1.1111110449e-01f,
                aT4 = 9.0908870101e-02f, aT5 = -7.6918758452e-02f, aT6 = 6.6610731184e-02f, aT7 = -5.8335702866e-02f,
                aT8 = 4.9768779427e-02f, aT9 = -3.6531571299e-02f, aT10 = 1.6285819933e-02f;
    const uint32_t hx = gfw_f2u(x);
    const uint32_t ix = hx & 0x7fffffffu;
    if (ix >= 0x4c000000u) {                    // |x| >= 2^25
        if (ix > 0x7f800000u) return x + x;     // NaN
        return (hx >> 31) ? -hi3 - lo3 : hi3 + lo3;
    }
    int id;
    float hi = 0.0f, lo = 0.0f;
    if (ix < 0x3ee00000u) {                     // |x| < 0.4375
        if (ix < 0x31000000u) return x;         // |x| < 2^-29
        id = -1;
    } else {
        x = gfw_fabsf(x);
        if (ix < 0x3f980000u) {                 // |x| < 1.1875
            if (ix < 0x3f300000u) { id = 0; hi = hi0; lo = lo0; x = (2.0f * x - 1.0f) / (2.0f + x); }
            else                  { id = 1; hi = hi1; lo = lo1; x = (x - 1.0f) / (x + 1.0f); }
        } else {
            if (ix < 0x401c0000u) { id = 2; hi = hi2; lo = lo2; x = (x - 1.5f) / (1.0f + 1.5f * x); }
            else                  { id = 3; hi = hi3; lo = lo3; x = -1.0f / x; }
        }
    }
    const float z = x * x;
    const float w = z * z;
    const float s1 = z * (aT0 + w * (aT2 + w * (aT4 + w * (aT6 + w * (aT8 + w * aT10)))));
    const float s2 = w * (aT1 + w * (aT3 + w * (aT5 + w * (aT7 + w * aT9))));
    if (id < 0) return x - x * (s1 + s2);
    const float r = hi - ((x * (s1 + s2) - lo) - x);
    return (hx >> 31) ? -r : r;
}

// ---- atanf for finite x >= 0 with the reduction read from a table ----------------------------------------------------
// The five reduced arguments of gfw_atanf are all (a_n*x + b_n) / (a_d*x + b_d) with the products exact or rounded exactly
// where the reference rounds (2x, 1.5x), so the interval only selects six constants: GFW_ATAN_TAB[id] = {a_n, b_n, a_d, b_d,
// hi, lo, 0, 0}.  id 0 (x < 0.4375) has hi = lo = 0 and t = x/1, and -(t*(s1+s2) - t) is x - x*(s1+s2) bit for bit; id 5
// (x >= 2^25) yields t = 0 and hi - (-lo) = hi3 + lo3.  Not for NaN / infinity (0 * inf).  The caller divides (IEEE `/` on the
// host, the lean correctly-rounded divide on the device); tests/test_math_host.py compares the composition with gfw_atanf on
// every finite non-negative float.
#define GFW_ATAN_TAB_INIT { \
    1.0f,  0.0f, 0.0f, 1.0f, 0.0f,             0.0f,             0.0f, 0.0f, \
    2.0f, -1.0f, 1.0f, 2.0f, 4.6364760399e-01f, 5.0121582440e-09f, 0.0f, 0.0f, \
    1.0f, -1.0f, 1.0f, 1.0f, 7.8539812565e-01f, 3.7748947079e-08f, 0.0f, 0.0f, \
    1.0f, -1.5f, 1.5f, 1.0f, 9.8279368877e-01f, 3.4473217170e-08f, 0.0f, 0.0f, \
    0.0f, -1.0f, 1.0f, 0.0f, 1.5707962513e+00f, 7.5497894159e-08f, 0.0f, 0.0f, \
    0.0f,  0.0f, 0.0f, 1.0f, 1.5707962513e+00f, 7.5497894159e-08f, 0.0f, 0.0f }
GFW_HD int gfw_atanf_tab_id(float x) {
    return (int)(x >= 0.4375f) + (int)(x >= 0.6875f) + (int)(x >= 1.1875f) + (int)(x >= 2.4375f) + (int)(x >= 33554432.0f);
}
GFW_HD void gfw_atanf_tab_reduce(float x, const float *rec, float *num, float *den) {
    const float pn = rec[0] * x, pd = rec[2] * x;       // separate roundings, as the reference's 2.0f*x and 1.5f*x
    *num = pn + rec[1];
    *den = pd + rec[3];
}
GFW_HD float gfw_atanf_tab_finish(float t, float hi, float lo) {
    const float aT0 = 3.3333334327e-01f, aT1 = -2.0000000298e-01f, aT2 = 1.4285714924e-01f, aT3 = -1.1111110449e-01f,
                aT4 = 9.0908870101e-02f, aT5 = -7.6918758452e-02f, aT6 = 6.6610731184e-02f, aT7 = -5.8335702866e-02f,
                aT8 = 4.9768779427e-02f, aT9 = -3.6531571299e-02f, aT10 = 1.6285819933e-02f;
    const float z = t * t;
    const float w = z * z;
    const float s1 = z * (aT0 + w * (aT2 + w * (aT4 + w * (aT6 + w * (aT8 + w * aT10)))));
    const float s2 = w * (aT1 + w * (aT3 + w * (aT5 + w * (aT7 + w * aT9))));
    return hi - ((t * (s1 + s2) - lo) - t);
}

// ---- tanf -------------------------------------------------------------------
// kernel on [-pi/4, pi/4] with a tail term y; iy = 1 -> tan, -1 -> -1/tan.
GFW_HD float gfw_kernel_tanf(float x, float y, int iy) {
    const float pio4 = 7.8539812565e-01f, pio4lo = 3.7748947079e-08f;
    const float T0 = 3.3333334327e-01f, T1 = 1.3333334029e-01f, T2 = 5.3968254477e-02f, T3 = 2.1869488060e-02f,
                T4 = 8.8632395491e-03f, T5 = 3.5920790397e-03f, T6 = 1.4562094584e-03f, T7 = 5.8804126456e-04f,
                T8 = 2.4646313977e-04f, T9 = 7.8179444245e-05f, T10 = 7.1407252108e-05f, T11 = -1.8558637748e-05f,
                T12 = 2.5907305826e-05f;
    const uint32_t hx = gfw_f2u(x);
    const uint32_t ix = hx & 0x7fffffffu;
    const int neg = (int)(hx >> 31);
    if (ix < 0x39000000u) {                     // |x| < 2^-13
        if ((int)x == 0) {
            if ((ix | (uint32_t)(iy + 1)) == 0) return 1.0f / gfw_fabsf(x);
            else if (iy == 1) return x;
            else return -1.0f / x;
        }
    }
    float z, w;
    if (ix >= 0x3f2ca140u) {                    // |x| >= 0.6744
        if (neg) { x = -x; y = -y; }
        z = pio4 - x;
        w = pio4lo - y;
        x = z + w; y = 0.0f;
        if (gfw_fabsf(x) < 0x1p-13f)
            return (float)(1 - (neg << 1)) * (float)iy * (1.0f - 2.0f * (float)iy * x);
    }
    z = x * x;
    w = z * z;
    float r = T1 + w * (T3 + w * (T5 + w * (T7 + w * (T9 + w * T11))));
    float v = z * (T2 + w * (T4 + w * (T6 + w * (T8 + w * (T10 + w * T12)))));
    float s = z * x;
    r = y + z * (s * (r + v) + y);
    r += T0 * s;
    w = x + r;
    if (ix >= 0x3f2ca140u) {
        v = (float)iy;
        return (float)(1 - (neg << 1)) * (v - 2.0f * (x - (w * w / (w + v) - r)));
    }
    if (iy == 1) return w;
    // -1/(x+r) to full precision
    z = gfw_u2f(gfw_f2u(w) & 0xfffff000u);
    v = r - (z - x);
    const float a = -1.0f / w;
    const float t = gfw_u2f(gfw_f2u(a) & 0xfffff000u);
    s = 1.0f + t * z;
    return t + a * (s + t * v);
}

// Range reduction as glibc >= 2.28 does it for tanf (e_rem_pio2f.c on top of the
// sincosf helpers of s_sincosf.h, Arm optimized-routines lineage): in double.
//   |x| < 120 : r = x * (2/pi * 2^24); n = ((int32)r + 2^23) >> 24; y = x - n * (pi/2)
//   otherwise : 32x96-bit fixed-point multiply by a 192-bit table of 4/pi.
// y0 = (float)y, y1 = (float)(y - y0).  Every double op is a single IEEE
// operation (no contraction), so host and gfx950 agree bit for bit.
GFW_HD int gfw_rem_pio2f(float x, float *y0, float *y1) {
    const uint32_t xi0 = gfw_f2u(x);
    double dx = (double)x;
    int n;
    if (((xi0 >> 20) & 0x7ffu) < 0x42fu) {        // abstop12(x) < abstop12(120.0f)
        const double hpi_inv = 0x1.45F306DC9C883p+23, hpi = 0x1.921FB54442D18p0;
        const double r = dx * hpi_inv;
        n = ((int32_t)r + 0x800000) >> 24;
        dx = dx - (double)n * hpi;
    } else {
        // 4/pi in 32-bit words with 8-bit overlap (the hexadecimal expansion of 2/pi)
        const uint32_t inv_pio4[24] = {
            0x000000a2u, 0x0000a2f9u, 0x00a2f983u, 0xa2f9836eu, 0xf9836e4eu, 0x836e4e44u, 0x6e4e4415u, 0x4e441529u,
            0x441529fcu, 0x1529fc27u, 0x29fc2757u, 0xfc2757d1u, 0x2757d1f5u, 0x57d1f534u, 0xd1f534ddu, 0xf534ddc0u,
            0x34ddc0dbu, 0xddc0db62u, 0xc0db6295u, 0xdb629599u, 0x6295993cu, 0x95993c43u, 0x993c4390u, 0x3c439041u};
        const double pi63 = 0x1.921FB54442D18p-62;
        uint32_t xi = xi0;
        const int sign = (int)(xi >> 31);
        const uint32_t *arr = &inv_pio4[(xi >> 26) & 15];
        const int shift = (int)((xi >> 23) & 7);
        xi = (xi & 0xffffffu) | 0x800000u;
        xi <<= shift;
        uint64_t res0 = (uint64_t)(uint32_t)(xi * arr[0]);
        const uint64_t res1 = (uint64_t)xi * arr[4];
        const uint64_t res2 = (uint64_t)xi * arr[8];
        res0 = (res2 >> 32) | (res0 << 32);
        res0 += res1;
        const uint64_t nn = (res0 + (1ULL << 61)) >> 62;
        res0 -= nn << 62;
        const double xr = (double)(int64_t)res0;
        n = (int)nn;
        dx = xr * pi63;
        dx = sign ? -dx : dx;
    }
    const float a = (float)dx;
    *y0 = a;
    *y1 = (float)(dx - (double)a);
    return n;
}

GFW_HD float gfw_tanf(float x) {
    const uint32_t ix = gfw_f2u(x) & 0x7fffffffu;
    if (ix <= 0x3f490fdau) return gfw_kernel_tanf(x, 0.0f, 1);
    if (ix >= 0x7f800000u) return x - x;
    float y0, y1;
    const int n = gfw_rem_pio2f(x, &y0, &y1);
    return gfw_kernel_tanf(y0, y1, 1 - ((n & 1) << 1));
}


// ---- sinf / cosf ----------------------------------------------------------------
// glibc >= 2.28 (sysdeps/ieee754/flt-32/s_sinf.c, s_cosf.c, s_sincosf.h — Arm optimized-routines lineage): the
// argument is reduced in double (the reduction gfw_rem_pio2f also restates) and a degree-7 / degree-8 double
// polynomial is evaluated, the result rounded to float once.  On x86-64 the dynamic linker selects the `_fma` build of
// these two routines on every FMA-capable CPU (sysdeps/x86_64/fpu/multiarch/s_sinf.c, s_cosf.c), in which the
// compiler contracted every `a + b * c` of the source into one fused operation; the sequence below is that build's,
// written with explicit fma() so that it does not depend on the contraction mode of THIS translation unit.
// tests/test_math_host.py compares it with the host libm on all 2^32 inputs; tests/test_gpu_math.py re-checks the
// device build.  Used for the IBIS/OIS roll angle (cpu_undistort.rs:159-160: cos(-m11), sin(-m11)) when the per-row
// matrices never visit the host.
#if defined(__HIP_DEVICE_COMPILE__)
#define GFW_FMA(a, b, c) __builtin_fma((a), (b), (c))
#else
#define GFW_FMA(a, b, c) __builtin_fma((a), (b), (c))
#endif
GFW_HD double gfw_sincosf_reduce(float y, int *np, int *flip_table) {
    // returns x*s (reduced argument with the quadrant sign applied); *np = quadrant; *flip_table = use the negated polynomial
    const uint32_t xi0 = gfw_f2u(y);
    const double sign4[4] = {1.0, -1.0, -1.0, 1.0};
    double x = (double)y;
    int n;
    if (((xi0 >> 20) & 0x7ffu) < 0x42fu) {           // abstop12(y) < abstop12(120.0f): reduce_fast
        const double hpi_inv = 0x1.45F306DC9C883p+23, hpi = 0x1.921FB54442D18p0;
        const double r = x * hpi_inv;
        n = ((int32_t)r + 0x800000) >> 24;
        x = GFW_FMA(-(double)n, hpi, x);
        *np = n;
        *flip_table = (n & 2) != 0;
        return x * sign4[n & 3];
    }
    // reduce_large
    const uint32_t inv_pio4[24] = {
        0x000000a2u, 0x0000a2f9u, 0x00a2f983u, 0xa2f9836eu, 0xf9836e4eu, 0x836e4e44u, 0x6e4e4415u, 0x4e441529u,
        0x441529fcu, 0x1529fc27u, 0x29fc2757u, 0xfc2757d1u, 0x2757d1f5u, 0x57d1f534u, 0xd1f534ddu, 0xf534ddc0u,
        0x34ddc0dbu, 0xddc0db62u, 0xc0db6295u, 0xdb629599u, 0x6295993cu, 0x95993c43u, 0x993c4390u, 0x3c439041u};
    const double pi63 = 0x1.921FB54442D18p-62;
    uint32_t xi = xi0;
    const int sign = (int)(xi >> 31);
    const uint32_t *arr = &inv_pio4[(xi >> 26) & 15];
    const int shift = (int)((xi >> 23) & 7);
    xi = (xi & 0xffffffu) | 0x800000u;
    xi <<= shift;
    uint64_t res0 = (uint64_t)(uint32_t)(xi * arr[0]);
    const uint64_t res1 = (uint64_t)xi * arr[4];
    const uint64_t res2 = (uint64_t)xi * arr[8];
    res0 = (res2 >> 32) | (res0 << 32);
    res0 += res1;
    const uint64_t nn = (res0 + (1ULL << 61)) >> 62;
    res0 -= nn << 62;
    x = (double)(int64_t)res0 * pi63;
    n = (int)nn;
    *np = n;
    *flip_table = ((n + sign) & 2) != 0;
    return x * sign4[(n + sign) & 3];
}
// sinf_poly of s_sincosf.h: n even -> sine polynomial, n odd -> cosine polynomial; `neg` selects __sincosf_table[1]
GFW_HD float gfw_sinf_poly(double x, double x2, int neg, int n) {
    const double c0 = 0x1p0, c1 = -0x1.ffffffd0c621cp-2, c2 = 0x1.55553e1068f19p-5, c3 = -0x1.6c087e89a359dp-10, c4 = 0x1.99343027bf8c3p-16;
    const double s1 = -0x1.555545995a603p-3, s2 = 0x1.1107605230bc4p-7, s3 = -0x1.994eb3774cf24p-13;
    if ((n & 1) == 0) {
        const double x3 = x * x2;
        const double t1 = GFW_FMA(x2, s3, s2);
        const double x7 = x3 * x2;
        const double s = GFW_FMA(x3, s1, x);
        return (float)GFW_FMA(x7, t1, s);
    }
    const double sg = neg ? -1.0 : 1.0;
    const double x4 = x2 * x2;
    const double d2 = GFW_FMA(x2, sg * c4, sg * c3);
    const double d1 = GFW_FMA(x2, sg * c1, sg * c0);
    const double x6 = x4 * x2;
    const double c = GFW_FMA(x4, sg * c2, d1);
    return (float)GFW_FMA(x6, d2, c);
}
GFW_HD float gfw_sinf(float y) {
    const uint32_t top = (gfw_f2u(y) >> 20) & 0x7ffu;
    if (top < 0x3f4u) {                                // abstop12(y) < abstop12(pio4 = 0x1.921FB6p-1f)
        if (top < 0x398u) return y;                    // |y| < 2^-12
        const double x = (double)y;
        return gfw_sinf_poly(x, x * x, 0, 0);
    }
    if (top >= 0x7f8u) return y - y;                   // inf / NaN
    int n, neg;
    const double xs = gfw_sincosf_reduce(y, &n, &neg);
    return gfw_sinf_poly(xs, xs * xs, neg, n);
}
GFW_HD float gfw_cosf(float y) {
    const uint32_t top = (gfw_f2u(y) >> 20) & 0x7ffu;
    if (top < 0x3f4u) {
        if (top < 0x398u) return 1.0f;
        const double x = (double)y;
        return gfw_sinf_poly(x, x * x, 0, 1);
    }
    if (top >= 0x7f8u) return y - y;
    int n, neg;
    const double xs = gfw_sincosf_reduce(y, &n, &neg);
    return gfw_sinf_poly(xs, xs * xs, neg, n ^ 1);
}
