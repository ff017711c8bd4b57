/* Test infrastructure — host runner for the reference's OWN OpenCL warp kernel (second opinion on the CPU side of the suite).
 *
 * oracle/build_ref_cl.py assembles src/core/gpu/opencl_undistort.cl + distortion_models/<model>.cl the way OclWrapper::new does
 * (opencl.rs:181-214) and, besides the gfx950 code objects, compiles the same assembled text with the ROCm clang for x86-64
 * (`-x cl --target=x86_64-unknown-linux-gnu -ffp-contract=off`).  That object leaves some forty OpenCL builtins undefined; this file defines
 * them (by their Itanium-mangled names, as clang's OpenCL front end emits them) with the semantics the OpenCL C specification gives
 * each — conversions: round toward zero, saturating, NaN -> 0 (6.2.3.3); min/max: `y < x ? y : x` / `x < y ? y : x` (6.12.4);
 * fmin/fmax/fabs/round/sin/cos/tan/atan/sqrt: the host libm — and provides the NDRange loop: gfw_ref_cl_run() calls
 * undistort_image() once per work-item with get_global_id() answering from thread-local storage.
 *
 * Nothing of the reference's text is in this file or in the repository; only the linked oracle/_ref/gfw_ref_cl_<name>.host.so
 * (git-ignored) holds its compiled form.  Like the gfx950 twin it is NOT golden: it is the reference's GPU backend, which deviates from
 * its CPU path where SURVEY.md section 8a says (sub-pixel rounding by convert_int_sat_rtz(0.5 + x), the r-limit test); on the host its
 * transcendental functions are glibc's, the same ones the reference's CPU path calls, which is what makes this build the tighter of
 * the two second opinions.  Only tests/ may load it.
 */
#include <math.h>
#include <stddef.h>
#include <stdint.h>

typedef float float2 __attribute__((ext_vector_type(2)));
typedef float float3 __attribute__((ext_vector_type(3)));
typedef float float4 __attribute__((ext_vector_type(4)));
typedef unsigned char uchar2 __attribute__((ext_vector_type(2)));
typedef unsigned char uchar4 __attribute__((ext_vector_type(4)));
typedef unsigned short ushort2 __attribute__((ext_vector_type(2)));
typedef unsigned short ushort4 __attribute__((ext_vector_type(4)));

#define OCL(ret, tag, mangled, ...) ret ocl_##tag(__VA_ARGS__) __asm__(mangled); ret ocl_##tag(__VA_ARGS__)

static __thread size_t g_id[3];

OCL(size_t, get_global_id, "_Z13get_global_idj", unsigned d) { return d < 3 ? g_id[d] : 0; }

/* explicit conversions, 6.2.3: float -> integer rounds toward zero by default; _sat clamps; NaN converts to 0 */
OCL(int, convert_int_sat_rtz, "_Z19convert_int_sat_rtzf", float x) {
    if (x != x) return 0;
    if (x >= 2147483648.0f) return INT32_MAX;
    if (x <= -2147483648.0f) return INT32_MIN;
    return (int)x;
}
OCL(unsigned short, convert_ushort_sat, "_Z18convert_ushort_satf", float x) {
    if (x != x || x <= 0.0f) return 0;
    if (x >= 65535.0f) return 65535;
    return (unsigned short)x;
}
OCL(unsigned char, convert_uchar_sat, "_Z17convert_uchar_satf", float x) {
    if (x != x || x <= 0.0f) return 0;
    if (x >= 255.0f) return 255;
    return (unsigned char)x;
}
OCL(float, convert_float_us, "_Z13convert_floatt", unsigned short v) { return (float)v; }
OCL(float, convert_float_uc, "_Z13convert_floath", unsigned char v) { return (float)v; }
OCL(float, convert_float_f, "_Z13convert_floatf", float v) { return v; }
OCL(float4, convert_float4, "_Z14convert_float4Dv4_f", float4 v) { return v; }
/* vector forms: component by component (6.2.3) */
OCL(uchar2, convert_uchar2_sat, "_Z18convert_uchar2_satDv2_f", float2 v) { return (uchar2){ocl_convert_uchar_sat(v.x), ocl_convert_uchar_sat(v.y)}; }
OCL(uchar4, convert_uchar4_sat, "_Z18convert_uchar4_satDv4_f", float4 v) {
    return (uchar4){ocl_convert_uchar_sat(v.x), ocl_convert_uchar_sat(v.y), ocl_convert_uchar_sat(v.z), ocl_convert_uchar_sat(v.w)};
}
OCL(ushort2, convert_ushort2_sat, "_Z19convert_ushort2_satDv2_f", float2 v) { return (ushort2){ocl_convert_ushort_sat(v.x), ocl_convert_ushort_sat(v.y)}; }
OCL(ushort4, convert_ushort4_sat, "_Z19convert_ushort4_satDv4_f", float4 v) {
    return (ushort4){ocl_convert_ushort_sat(v.x), ocl_convert_ushort_sat(v.y), ocl_convert_ushort_sat(v.z), ocl_convert_ushort_sat(v.w)};
}
OCL(float2, convert_float2_uc, "_Z14convert_float2Dv2_h", uchar2 v) { return (float2){(float)v.x, (float)v.y}; }
OCL(float2, convert_float2_us, "_Z14convert_float2Dv2_t", ushort2 v) { return (float2){(float)v.x, (float)v.y}; }
OCL(float4, convert_float4_uc, "_Z14convert_float4Dv4_h", uchar4 v) { return (float4){(float)v.x, (float)v.y, (float)v.z, (float)v.w}; }
OCL(float4, convert_float4_us, "_Z14convert_float4Dv4_t", ushort4 v) { return (float4){(float)v.x, (float)v.y, (float)v.z, (float)v.w}; }

/* cl_khr_fp16 loads / stores (6.12.7 vload_half / vstore_half_rte): IEEE binary16 storage, round to nearest even on the way in */
OCL(float4, vload_half4, "_Z11vload_half4mPU9CLgenericKDh", size_t offset, const _Float16 *p) {
    p += 4 * offset;
    return (float4){(float)p[0], (float)p[1], (float)p[2], (float)p[3]};
}
OCL(void, vstore_half4_rte, "_Z16vstore_half4_rteDv4_fmPU9CLgenericDh", float4 v, size_t offset, _Float16 *p) {
    p += 4 * offset;
    p[0] = (_Float16)v.x; p[1] = (_Float16)v.y; p[2] = (_Float16)v.z; p[3] = (_Float16)v.w;
}

/* common functions, 6.12.4 */
static inline float min_f(float x, float y) { return y < x ? y : x; }
static inline float max_f(float x, float y) { return x < y ? y : x; }
OCL(float, min_ff, "_Z3minff", float x, float y) { return min_f(x, y); }
OCL(float, max_ff, "_Z3maxff", float x, float y) { return max_f(x, y); }
OCL(float, clamp_fff, "_Z5clampfff", float x, float lo, float hi) { return min_f(max_f(x, lo), hi); }      /* min(max(x, minval), maxval) */
OCL(int, min_ii, "_Z3minii", int x, int y) { return y < x ? y : x; }
OCL(int, max_ii, "_Z3maxii", int x, int y) { return x < y ? y : x; }
OCL(float2, min_v2, "_Z3minDv2_fS_", float2 x, float2 y) { return (float2){min_f(x.x, y.x), min_f(x.y, y.y)}; }
OCL(float2, max_v2, "_Z3maxDv2_fS_", float2 x, float2 y) { return (float2){max_f(x.x, y.x), max_f(x.y, y.y)}; }
OCL(float4, min_v4, "_Z3minDv4_fS_", float4 x, float4 y) { return (float4){min_f(x.x, y.x), min_f(x.y, y.y), min_f(x.z, y.z), min_f(x.w, y.w)}; }

/* math functions, 6.12.2: the host libm (glibc: what the reference's CPU path calls through Rust's f32 methods) */
OCL(float, fmin_ff, "_Z4fminff", float x, float y) { return fminf(x, y); }
OCL(float, fmax_ff, "_Z4fmaxff", float x, float y) { return fmaxf(x, y); }
OCL(float, fabs_f, "_Z4fabsf", float x) { return fabsf(x); }
OCL(float2, fabs_v2, "_Z4fabsDv2_f", float2 x) { return (float2){fabsf(x.x), fabsf(x.y)}; }
OCL(float, round_f, "_Z5roundf", float x) { return roundf(x); }
OCL(float2, round_v2, "_Z5roundDv2_f", float2 x) { return (float2){roundf(x.x), roundf(x.y)}; }
OCL(float, sin_f, "_Z3sinf", float x) { return sinf(x); }
OCL(float, cos_f, "_Z3cosf", float x) { return cosf(x); }
OCL(float, tan_f, "_Z3tanf", float x) { return tanf(x); }
OCL(float, atan_f, "_Z4atanf", float x) { return atanf(x); }
OCL(float, sqrt_f, "_Z4sqrtf", float x) { return sqrtf(x); }
OCL(float, floor_f, "_Z5floorf", float x) { return floorf(x); }
OCL(float, ceil_f, "_Z4ceilf", float x) { return ceilf(x); }
/* geometric functions, 6.12.5 */
OCL(float, length_v2, "_Z6lengthDv2_f", float2 v) { return sqrtf(v.x * v.x + v.y * v.y); }
OCL(float, length_v3, "_Z6lengthDv3_f", float3 v) { return sqrtf(v.x * v.x + v.y * v.y + v.z * v.z); }

/* the reference's kernel (opencl_undistort.cl:582), compiled from its own source into the object this file is linked with */
extern void undistort_image(const unsigned char *srcptr, unsigned char *dstptr, const void *params_buf, const float *matrices,
                            const unsigned char *drawing, const float *mesh_data);

/* One work-item per (x, y) of a global_w x global_h NDRange, rows y0 <= y < y1 (the caller may split rows over threads). */
void gfw_ref_cl_run(const unsigned char *src, unsigned char *dst, const void *params, const float *matrices, const unsigned char *drawing,
                    const float *mesh, int global_w, int y0, int y1) {
    g_id[2] = 0;
    for (int y = y0; y < y1; ++y) {
        g_id[1] = (size_t)y;
        for (int x = 0; x < global_w; ++x) {
            g_id[0] = (size_t)x;
            undistort_image(src, dst, params, matrices, drawing, mesh);
        }
    }
}
