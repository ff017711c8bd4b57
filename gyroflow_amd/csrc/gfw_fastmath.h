// gfw_fastmath.h — correctly-rounded f32 divide / sqrt / atan for the hot kernel, without the range
// scaffolding of the generic expansions.
//
// The generic IEEE `a / b` and `sqrtf` that hipcc emits carry v_div_scale / v_div_fixup / 2^32 pre-scaling so
// that they are right for denormal and near-overflow operands (measured on MI355X: ~19 and ~23 plain-VALU
// issue slots each, tools/microbench.hip).  The warp kernel's operands live in a narrow, checked range, so
// the same Newton/FMA refinement can run without that scaffolding.  Every routine here returns exactly the
// correctly rounded (round-to-nearest-even) result — i.e. the same bits as the generic expansion — for
// operands inside the stated range; callers check the range once per pixel and take the generic path
// otherwise.  tests/test_gpu_math.py compares each routine with the generic one on ~10^9 operands.
#pragma once
#include <hip/hip_runtime.h>
#include "gfw_math.h"

// 1-ulp hardware approximations
__device__ __forceinline__ float gfw_hw_rcp(float x) { return __builtin_amdgcn_rcpf(x); }
__device__ __forceinline__ float gfw_hw_sqrt(float x) { return __builtin_amdgcn_sqrtf(x); }

// Reciprocal refined to < 1/2 ulp + tiny, shared by every quotient with the same denominator.
struct GfwRcp { float b, r; };
__device__ __forceinline__ GfwRcp gfw_rcp_prepare(float b) {
    const float r0 = gfw_hw_rcp(b);
    const float e = __builtin_fmaf(-b, r0, 1.0f);
    return GfwRcp{b, __builtin_fmaf(e, r0, r0)};
}
// RN(a / b) for |b| in [2^-60, 2^60], |a| <= 2^60 (a == 0 allowed; the sign of a zero result may differ).
// Refined reciprocal, q0 = a*r, ONE exact-remainder correction (Markstein's scheme).  That this is the correctly
// rounded quotient with MI355X's v_rcp_f32 is machine-checked, not assumed: tools/prove_div.py compares it with the
// generic expansion for all 2^23 x 2^23 = 7.04e13 significand pairs on the device (0 mismatches,
// profiles/r01_div_one_correction_proof.txt; with the raw, unrefined v_rcp the same scheme does fail), and every
// operation scales exactly with the operands' exponents inside the stated range.
__device__ __forceinline__ float gfw_div_prepared(float a, const GfwRcp &d) {
    const float q0 = a * d.r;
    const float r0 = __builtin_fmaf(-d.b, q0, a);
    return __builtin_fmaf(r0, d.r, q0);
}
__device__ __forceinline__ float gfw_div_lean(float a, float b) { return gfw_div_prepared(a, gfw_rcp_prepare(b)); }

// RN(sqrt(x)) for x == +-0 or x in [2^-80, 2^80]: hardware sqrt (<= 1 ulp) then pick between s-1ulp, s, s+1ulp by
// the sign of the exact residuals x - s_down*s and x - s_up*s (the test the generic lowering uses).  Zero needs no case of its own:
// s = +-0, its lower neighbour's bit pattern is a NaN (both comparisons fail) and the upper neighbour's residual is a signed zero.
__device__ __forceinline__ float gfw_sqrt_lean(float x) {
    const float s = gfw_hw_sqrt(x);
    const float s_dn = gfw_u2f(gfw_f2u(s) - 1u);
    const float s_up = gfw_u2f(gfw_f2u(s) + 1u);
    const float vp = __builtin_fmaf(-s_dn, s, x);
    const float vs = __builtin_fmaf(-s_up, s, x);
    float r = (vp <= 0.0f) ? s_dn : s;
    r = (vs > 0.0f) ? s_up : r;
    return r;
}

// glibc-2.35 atanf (gfw_math.h: gfw_atanf) for x >= 0 (or NaN), select-based: one division, no branches.
//   id -1: x < 0.4375          -> t = x
//   id  0: x < 0.6875          -> t = (2x-1)/(2+x)
//   id  1: x < 1.1875          -> t = (x-1)/(x+1)
//   id  2: x < 2.4375          -> t = (x-1.5)/(1+1.5x)
//   id  3: otherwise           -> t = -1/x          (x >= 2^25 returns RN(pi/2) like the reference)
__device__ __forceinline__ float gfw_atanf_pos(float x) {
    const float aT0 = 3.3333334327e-01f, aT1 = -2.0000000298e-01f, aT2 = 1.4285714924e-01f, aT3 = -1.1111110449e-01f,
                aT4 = 9.0908870101e-02f, aT5 = -7.6918758452e-02f, aT6 = 6.6610731184e-02f, aT7 = -5.8335702866e-02f,
                aT8 = 4.9768779427e-02f, aT9 = -3.6531571299e-02f, aT10 = 1.6285819933e-02f;
    const bool lt0 = x < 0.4375f, lt1 = x < 0.6875f, lt2 = x < 1.1875f, lt3 = x < 2.4375f;
    // numerator / denominator of the reduced argument
    const float n01 = lt1 ? (2.0f * x - 1.0f) : (x - 1.0f);
    const float d01 = lt1 ? (2.0f + x) : (x + 1.0f);
    const float n23 = lt3 ? (x - 1.5f) : -1.0f;
    const float d23 = lt3 ? (1.0f + 1.5f * x) : x;
    float num = lt2 ? n01 : n23;
    float den = lt2 ? d01 : d23;
    num = lt0 ? x : num;
    den = lt0 ? 1.0f : den;
    const float hi = lt2 ? (lt1 ? 4.6364760399e-01f : 7.8539812565e-01f) : (lt3 ? 9.8279368877e-01f : 1.5707962513e+00f);
    const float lo = lt2 ? (lt1 ? 5.0121582440e-09f : 3.7748947079e-08f) : (lt3 ? 3.4473217170e-08f : 7.5497894159e-08f);
    const float t = gfw_div_lean(num, den);     // den in [1, 2^25), |num| in {0} U [2^-24, 2^25): always in range
    const float z = t * t;
    const float w = z * z;
    const float s1 = z * (aT0 + w * (aT2 + w * (aT4 + w * (aT6 + w * (aT8 + w * aT10)))));
    const float s2 = w * (aT1 + w * (aT3 + w * (aT5 + w * (aT7 + w * aT9))));
    const float ts = t * (s1 + s2);
    const float r_small = t - ts;
    const float r_big = hi - ((ts - lo) - t);
    float r = lt0 ? r_small : r_big;
    r = (x >= 33554432.0f) ? 1.5707963705062866f : r;    // |x| >= 2^25: atanhi[3] + atanlo[3]
    return (x != x) ? x + x : r;
}

// The same function with the reduction read from a 6-record table in LDS instead of selected (gfw_math.h: gfw_atanf_tab_*;
// host-checked against gfw_atanf on every finite float >= 0).  For FINITE x >= 0 only (0 * inf would be NaN): the fused kernel
// calls it on r = sqrt(a^2 + b^2) of operands inside the lean range, which is finite.  ~17 VALU instructions fewer than the
// select-based form, two LDS reads more.  Opt-in (GFW_ATAN_TABLE) until it has been timed on the device.
__device__ __forceinline__ float *gfw_atan_lds() { __shared__ float tab[48]; return tab; }
__device__ const float GFW_ATAN_TAB[48] = GFW_ATAN_TAB_INIT;
__device__ __forceinline__ void gfw_atan_lds_init(int tid) {                       // every thread of the workgroup, before the first use
    if (tid < 48) gfw_atan_lds()[tid] = GFW_ATAN_TAB[tid];
    __syncthreads();
}
__device__ __forceinline__ float gfw_atanf_pos_tab(float x) {
    const float *rec = gfw_atan_lds() + 8 * gfw_atanf_tab_id(x);
    const float4 ab = *reinterpret_cast<const float4 *>(rec);
    const float2 hl = *reinterpret_cast<const float2 *>(rec + 4);
    const float pn = ab.x * x, pd = ab.z * x;
    const float num = pn + ab.y, den = pd + ab.w;
    return gfw_atanf_tab_finish(gfw_div_lean(num, den), hl.x, hl.y);           // den in [1, 2^25.6], |num| <= 2^26: in range
}

// The record index without the five compares (round 4): for a positive float the reduction interval is a step function of the bit pattern, and
// every threshold but the last is a multiple of 2^18 in it (0.4375 = 0x3ee00000, 0.6875 = 0x3f300000, 1.1875 = 0x3f980000, 2.4375 = 0x401c0000), so
// key = clamp((bits >> 18) - 0xfb7, 0, 80) names the interval and an 81-byte LDS table turns it into the record's offset: three VALU
// instructions and a byte read where the compare / select chain took ten and five VCC hazards.  x >= 2^25 (record 5) is NOT representable by
// the key: callers keep such operands away (the branch-free row flags them as odd; rr < 2^50 there).
#define GFW_ATAN_KEYS 81
__device__ __forceinline__ unsigned char *gfw_atan_key_lds() { __shared__ unsigned char lut[GFW_ATAN_KEYS + 3]; return lut; }
__device__ __forceinline__ void gfw_atan_key_lds_init(int tid) {                   // every thread of the workgroup, before the first use
    if (tid < GFW_ATAN_KEYS) gfw_atan_key_lds()[tid] = (unsigned char)(gfw_atanf_tab_id(gfw_u2f((uint32_t)(tid + 0xfb7) << 18)) * 32);
    __syncthreads();
}
__device__ __forceinline__ float gfw_atanf_pos_key(float x) {                      // finite x in [0, 2^25) (anything else: some record, garbage out, no fault)
    int key = (int)(gfw_f2u(x) >> 18) - 0xfb7;
    key = max(min(key, GFW_ATAN_KEYS - 1), 0);
    const float *rec = reinterpret_cast<const float *>(reinterpret_cast<const unsigned char *>(gfw_atan_lds()) + gfw_atan_key_lds()[key]);
    const float4 ab = *reinterpret_cast<const float4 *>(rec);
    const float2 hl = *reinterpret_cast<const float2 *>(rec + 4);
    const float pn = ab.x * x, pd = ab.z * x;
    const float num = pn + ab.y, den = pd + ab.w;
    return gfw_atanf_tab_finish(gfw_div_lean(num, den), hl.x, hl.y);
}

// RN(RN(x * mul) / den) for a constant (mul, den) pair: the reference's map_coord with in_min = out_min = 0,
//   (x - 0) * (out_max - 0) / (in_max - 0) + 0            (util.rs:144-147)
// evaluated with the precomputed RN(1/den).  The host validates (exhaustively over all 2^23 significands) that
// the two-FMA correction yields the correctly rounded quotient for this `den` before a kernel may use it.
struct GfwMapConst { float mul, den, rcp; };
__device__ __forceinline__ float gfw_map_const(float x, const GfwMapConst &m) {
    const float a = x * m.mul;
    const float q0 = a * m.rcp;
    const float r0 = __builtin_fmaf(-m.den, q0, a);
    return __builtin_fmaf(r0, m.rcp, q0);
}
