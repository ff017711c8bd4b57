// tbuffer_probe.hip — can the memory pipeline do the taps' integer -> float conversion?  (design data for the LUT samplers; not part of the product)
// Round-5 verdict, weak #3: a third of a Lanczos4 tap row is v_cvt_f32_u32 (SDWA) + v_alignbit on the VALU.  A typed buffer load
// (tbuffer_load_format_xyzw, BUF_DATA_FORMAT_16_16_16_16 / 8_8_8_8, BUF_NUM_FORMAT_USCALED) returns the samples as floats, exact for <= 16-bit integers.
// This probe checks (1) that the values are exact at every 2-byte (u16) / 1-byte (u8) alignment, (2) what a tap row costs either way on a 4K plane walked the
// way the fused kernel walks it (128 x 16-pixel tiles, a pixel pair per lane, 8 x 8 taps per sample), with the arithmetic of the reference (mul, add per tap; no fma)
// and without it (fetch only: the memory pipeline's own rate).
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off tools/tbuffer_probe.hip -o tools/tbuffer_probe
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
typedef int i4 __attribute__((ext_vector_type(4)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f2 __attribute__((ext_vector_type(2)));
__device__ f4 tbuf_load4(i4 rsrc, int voffset, int soffset, int format, int aux) __asm("llvm.amdgcn.raw.tbuffer.load.v4f32");
__device__ f2 tbuf_load2(i4 rsrc, int voffset, int soffset, int format, int aux) __asm("llvm.amdgcn.raw.tbuffer.load.v2f32");
__device__ float tbuf_load1(i4 rsrc, int voffset, int soffset, int format, int aux) __asm("llvm.amdgcn.raw.tbuffer.load.f32");
// gfx9 MTBUF format immediate: dfmt | nfmt << 4
constexpr int DF_8 = 1, DF_16 = 2, DF_8_8 = 3, DF_16_16 = 5, DF_8_8_8_8 = 10, DF_16_16_16_16 = 12, NF_USCALED = 2, NF_UINT = 4;
#define FMT(d, n) ((d) | ((n) << 4))

__device__ __forceinline__ i4 make_rsrc(const void *p, uint32_t bytes, uint32_t dword3) {
    const uint64_t a = (uint64_t)p;
    i4 r; r.x = (int)(uint32_t)a; r.y = (int)(uint32_t)((a >> 32) & 0xffffu); r.z = (int)bytes; r.w = (int)dword3;
    r.x = __builtin_amdgcn_readfirstlane(r.x); r.y = __builtin_amdgcn_readfirstlane(r.y); r.z = __builtin_amdgcn_readfirstlane(r.z); r.w = __builtin_amdgcn_readfirstlane(r.w);
    return r;
}
constexpr uint32_t RSRC3 = 0x00020FACu;          // dst_sel x,y,z,w = R,G,B,A (4,5,6,7); data_format 32 (ignored by MTBUF: the instruction carries its own)

// ---- (1) exactness at every alignment -----------------------------------------------------------------------------------------------------------------
__global__ void k_check(const uint16_t *s16, const uint8_t *s8, uint32_t n16, uint32_t n8, unsigned *bad, uint32_t rsrc3) {
    const uint32_t e = blockIdx.x * blockDim.x + threadIdx.x;          // element index: every alignment occurs
    const i4 r16 = make_rsrc(s16, n16 * 2, rsrc3), r8 = make_rsrc(s8, n8, rsrc3);
    if (e + 4 <= n16) {
        const f4 a = tbuf_load4(r16, (int)(e * 2), 0, FMT(DF_16_16_16_16, NF_USCALED), 0);
        const f2 b = tbuf_load2(r16, (int)(e * 2), 0, FMT(DF_16_16, NF_USCALED), 0);
        const float c = tbuf_load1(r16, (int)(e * 2), 0, FMT(DF_16, NF_USCALED), 0);
        if (a.x != (float)s16[e] || a.y != (float)s16[e + 1] || a.z != (float)s16[e + 2] || a.w != (float)s16[e + 3]) atomicAdd(&bad[0], 1u);
        if (b.x != (float)s16[e] || b.y != (float)s16[e + 1]) atomicAdd(&bad[1], 1u);
        if (c != (float)s16[e]) atomicAdd(&bad[2], 1u);
    }
    if (e + 4 <= n8) {
        const f4 a = tbuf_load4(r8, (int)e, 0, FMT(DF_8_8_8_8, NF_USCALED), 0);
        const f2 b = tbuf_load2(r8, (int)e, 0, FMT(DF_8_8, NF_USCALED), 0);
        const float c = tbuf_load1(r8, (int)e, 0, FMT(DF_8, NF_USCALED), 0);
        if (a.x != (float)s8[e] || a.y != (float)s8[e + 1] || a.z != (float)s8[e + 2] || a.w != (float)s8[e + 3]) atomicAdd(&bad[3], 1u);
        if (b.x != (float)s8[e] || b.y != (float)s8[e + 1]) atomicAdd(&bad[4], 1u);
        if (c != (float)s8[e]) atomicAdd(&bad[5], 1u);
    }
}

// ---- (2) a tap block either way ----------------------------------------------------------------------------------------------------------------------
// MODE 0: aligned dword fetches + v_alignbit + v_cvt (today's taps_inside)   MODE 1: typed loads
// ARITH 1: the reference's row-then-column sums (mul, add per tap)           ARITH 0: fetch only (the values are xor-ed together)
// I: taps per row and rows per sample (8 Lanczos4, 4 bicubic, 2 bilinear)
template <typename T, int I, int MODE, int ARITH>
__global__ __launch_bounds__(256) void k_taps(const uint8_t *src, int stride, int w, int h, float *out, const float *lut, int shift_x) {
    __shared__ float s_lut[448];
    for (int i = threadIdx.x + threadIdx.y * 64; i < 448; i += 256) s_lut[i] = lut[i];
    __syncthreads();
    const int tiles_x = w / 128;
    const int tx = blockIdx.x % tiles_x, ty = blockIdx.x / tiles_x;
    const int lane = threadIdx.x, wave = threadIdx.y;
    const i4 rs = make_rsrc(src, (uint32_t)stride * (uint32_t)h, RSRC3);
    float acc = 0.0f; uint32_t xacc = 0;
    #pragma unroll 1
    for (int r = 0; r < 4; ++r) {
        const int y = ty * 16 + wave * 4 + r;
        #pragma unroll 1
        for (int px = 0; px < 2; ++px) {
            const int x = tx * 128 + lane * 2 + px;
            int sx = x + shift_x + ((y >> 2) & 1), sy = y + ((lane >> 4) & 1);           // a smooth, slightly sheared map: both parities of sx occur in every wave
            sx = min(max(sx, 0), w - I - 4); sy = min(max(sy, 0), h - I);
            const float *cx = s_lut + 192 + ((x * 5) & 31) * 8, *cy = s_lut + 192 + ((y * 3) & 31) * 8;
            const uint32_t off0 = (uint32_t)sy * (uint32_t)stride + (uint32_t)sx * (uint32_t)sizeof(T);
            float s1 = 0.0f;
            if (MODE == 0) {
                constexpr int ND = (I * (int)sizeof(T)) / 4 > 0 ? (I * (int)sizeof(T)) / 4 : 1;
                const unsigned mis = off0 & 3u, sh = mis * 8u;
                uint32_t aoff = off0 & ~3u;
                #pragma unroll
                for (int yp = 0; yp < I; ++yp) {
                    const uint32_t *wp = reinterpret_cast<const uint32_t *>(src + aoff);
                    uint32_t wd[ND + 1];
                    #pragma unroll
                    for (int j = 0; j < ND + 1; ++j) wd[j] = wp[j];
                    float xs = 0.0f;
                    #pragma unroll
                    for (int j = 0; j < ND; ++j) {
                        const uint32_t d = __builtin_amdgcn_alignbit(wd[j + 1], wd[j], sh);
                        if (ARITH) {
                            if (sizeof(T) == 2) {
                                const float t0 = (float)(d & 0xffffu) * cx[2 * j];
                                xs = (j == 0) ? t0 : xs + t0;
                                if (2 * j + 1 < I) xs = xs + (float)(d >> 16) * cx[2 * j + 1];
                            } else {
                                const float t0 = (float)(d & 0xffu) * cx[4 * j];
                                xs = (j == 0) ? t0 : xs + t0;
                                if (4 * j + 1 < I) xs = xs + (float)((d >> 8) & 0xffu) * cx[4 * j + 1];
                                if (4 * j + 2 < I) xs = xs + (float)((d >> 16) & 0xffu) * cx[4 * j + 2];
                                if (4 * j + 3 < I) xs = xs + (float)(d >> 24) * cx[4 * j + 3];
                            }
                        } else xacc ^= d;
                    }
                    if (ARITH) s1 = s1 + xs * cy[yp];
                    aoff += (uint32_t)stride;
                }
            } else {
                uint32_t o = off0;
                #pragma unroll
                for (int yp = 0; yp < I; ++yp) {
                    float t[8];
                    if (I == 8) {
                        const f4 a = tbuf_load4(rs, (int)o, 0, sizeof(T) == 2 ? FMT(DF_16_16_16_16, NF_USCALED) : FMT(DF_8_8_8_8, NF_USCALED), 0);
                        const f4 b = tbuf_load4(rs, (int)(o + 4 * sizeof(T)), 0, sizeof(T) == 2 ? FMT(DF_16_16_16_16, NF_USCALED) : FMT(DF_8_8_8_8, NF_USCALED), 0);
                        t[0] = a.x; t[1] = a.y; t[2] = a.z; t[3] = a.w; t[4] = b.x; t[5] = b.y; t[6] = b.z; t[7] = b.w;
                    } else if (I == 4) {
                        const f4 a = tbuf_load4(rs, (int)o, 0, sizeof(T) == 2 ? FMT(DF_16_16_16_16, NF_USCALED) : FMT(DF_8_8_8_8, NF_USCALED), 0);
                        t[0] = a.x; t[1] = a.y; t[2] = a.z; t[3] = a.w;
                    } else {
                        const f2 a = tbuf_load2(rs, (int)o, 0, sizeof(T) == 2 ? FMT(DF_16_16, NF_USCALED) : FMT(DF_8_8, NF_USCALED), 0);
                        t[0] = a.x; t[1] = a.y;
                    }
                    if (ARITH) {
                        float xs = t[0] * cx[0];
                        #pragma unroll
                        for (int j = 1; j < I; ++j) xs = xs + t[j] * cx[j];
                        s1 = s1 + xs * cy[yp];
                    } else {
                        #pragma unroll
                        for (int j = 0; j < I; ++j) xacc ^= __builtin_bit_cast(uint32_t, t[j]);
                    }
                    o += (uint32_t)stride;
                }
            }
            acc += s1;
        }
    }
    out[(size_t)blockIdx.x * 256 + wave * 64 + lane] = acc + (ARITH ? 0.0f : (float)(xacc & 0xffffu));
}

template <typename T, int I, int MODE, int ARITH>
static double run(const char *name, const uint8_t *d_src, int stride, int w, int h, float *d_out, const float *d_lut, std::vector<float> *res) {
    const int grid = (w / 128) * (h / 16);
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((k_taps<T, I, MODE, ARITH>), dim3(grid), dim3(64, 4), 0, 0, d_src, stride, w, h, d_out, d_lut, 3);
    CHECK(hipDeviceSynchronize());
    const int reps = 20;
    CHECK(hipEventRecord(e0));
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((k_taps<T, I, MODE, ARITH>), dim3(grid), dim3(64, 4), 0, 0, d_src, stride, w, h, d_out, d_lut, 3);
    CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    if (res) { res->resize((size_t)grid * 256); CHECK(hipMemcpy(res->data(), d_out, res->size() * 4, hipMemcpyDeviceToHost)); }
    printf("  %-64s %8.2f us per plane pass\n", name, ms * 1e3 / reps);
    return ms * 1e3 / reps;
}

int main() {
    const int w = 3840, h = 2160;
    std::vector<uint16_t> h16((size_t)w * h); std::vector<uint8_t> h8((size_t)w * h);
    uint32_t s = 0x9F10u;
    for (size_t i = 0; i < h16.size(); ++i) { s = s * 1664525u + 1013904223u; h16[i] = (uint16_t)(s >> 16); h8[i] = (uint8_t)(s >> 8); }
    h16[0] = 65535; h16[1] = 0; h16[2] = 32768; h8[0] = 255; h8[1] = 0;
    uint16_t *d16; uint8_t *d8; unsigned *d_bad; float *d_out, *d_lut;
    CHECK(hipMalloc(&d16, h16.size() * 2 + 64)); CHECK(hipMalloc(&d8, h8.size() + 64)); CHECK(hipMalloc(&d_bad, 64)); CHECK(hipMalloc(&d_out, (size_t)(w / 128) * (h / 16) * 256 * 4)); CHECK(hipMalloc(&d_lut, 448 * 4));
    CHECK(hipMemcpy(d16, h16.data(), h16.size() * 2, hipMemcpyHostToDevice)); CHECK(hipMemcpy(d8, h8.data(), h8.size(), hipMemcpyHostToDevice));
    std::vector<float> lut(448); for (int i = 0; i < 448; ++i) lut[i] = 0.001f * (float)((i * 37) % 211) - 0.05f;
    CHECK(hipMemcpy(d_lut, lut.data(), 448 * 4, hipMemcpyHostToDevice));
    for (uint32_t rsrc3 : {RSRC3, 0x00020000u}) {
        CHECK(hipMemset(d_bad, 0, 64));
        const uint32_t n = 1u << 20;
        hipLaunchKernelGGL(k_check, dim3(n / 256), dim3(256), 0, 0, d16, d8, n, n, d_bad, rsrc3);
        unsigned bad[6]; CHECK(hipMemcpy(bad, d_bad, sizeof(bad), hipMemcpyDeviceToHost));
        printf("typed loads, USCALED, every alignment over 2^20 elements, resource dword3 0x%08x: mismatches 16x4 %u, 16x2 %u, 16 %u, 8x4 %u, 8x2 %u, 8 %u\n", rsrc3, bad[0], bad[1], bad[2], bad[3], bad[4], bad[5]);
    }
    std::vector<float> a, b;
    printf("u16 plane 3840 x 2160, one sample per pixel:\n");
    run<uint16_t, 8, 0, 1>("Lanczos4 8x8, aligned dwords + alignbit + cvt, mul/add", (const uint8_t *)d16, w * 2, w, h, d_out, d_lut, &a);
    run<uint16_t, 8, 1, 1>("Lanczos4 8x8, typed 16_16_16_16 x2 per row, mul/add", (const uint8_t *)d16, w * 2, w, h, d_out, d_lut, &b);
    printf("    sums bit-identical: %s\n", (a.size() == b.size() && memcmp(a.data(), b.data(), a.size() * 4) == 0) ? "yes" : "NO");
    run<uint16_t, 8, 0, 0>("Lanczos4 8x8, aligned dwords + alignbit, fetch only", (const uint8_t *)d16, w * 2, w, h, d_out, d_lut, nullptr);
    run<uint16_t, 8, 1, 0>("Lanczos4 8x8, typed loads, fetch only", (const uint8_t *)d16, w * 2, w, h, d_out, d_lut, nullptr);
    run<uint16_t, 4, 0, 1>("bicubic 4x4, aligned dwords + alignbit + cvt, mul/add", (const uint8_t *)d16, w * 2, w, h, d_out, d_lut, &a);
    run<uint16_t, 4, 1, 1>("bicubic 4x4, typed 16_16_16_16 per row, mul/add", (const uint8_t *)d16, w * 2, w, h, d_out, d_lut, &b);
    printf("    sums bit-identical: %s\n", (a.size() == b.size() && memcmp(a.data(), b.data(), a.size() * 4) == 0) ? "yes" : "NO");
    run<uint16_t, 4, 0, 0>("bicubic 4x4, aligned dwords, fetch only", (const uint8_t *)d16, w * 2, w, h, d_out, d_lut, nullptr);
    run<uint16_t, 4, 1, 0>("bicubic 4x4, typed loads, fetch only", (const uint8_t *)d16, w * 2, w, h, d_out, d_lut, nullptr);
    run<uint16_t, 2, 0, 1>("bilinear 2x2, aligned dwords + alignbit + cvt, mul/add", (const uint8_t *)d16, w * 2, w, h, d_out, d_lut, &a);
    run<uint16_t, 2, 1, 1>("bilinear 2x2, typed 16_16 per row, mul/add", (const uint8_t *)d16, w * 2, w, h, d_out, d_lut, &b);
    printf("    sums bit-identical: %s\n", (a.size() == b.size() && memcmp(a.data(), b.data(), a.size() * 4) == 0) ? "yes" : "NO");
    printf("u8 plane 3840 x 2160:\n");
    run<uint8_t, 8, 0, 1>("Lanczos4 8x8, aligned dwords + alignbit + cvt_ubyte, mul/add", d8, w, w, h, d_out, d_lut, &a);
    run<uint8_t, 8, 1, 1>("Lanczos4 8x8, typed 8_8_8_8 x2 per row, mul/add", d8, w, w, h, d_out, d_lut, &b);
    printf("    sums bit-identical: %s\n", (a.size() == b.size() && memcmp(a.data(), b.data(), a.size() * 4) == 0) ? "yes" : "NO");
    run<uint8_t, 8, 0, 0>("Lanczos4 8x8, aligned dwords, fetch only", d8, w, w, h, d_out, d_lut, nullptr);
    run<uint8_t, 8, 1, 0>("Lanczos4 8x8, typed loads, fetch only", d8, w, w, h, d_out, d_lut, nullptr);
    run<uint8_t, 4, 0, 1>("bicubic 4x4, aligned dwords + alignbit + cvt_ubyte, mul/add", d8, w, w, h, d_out, d_lut, &a);
    run<uint8_t, 4, 1, 1>("bicubic 4x4, typed 8_8_8_8 per row, mul/add", d8, w, w, h, d_out, d_lut, &b);
    printf("    sums bit-identical: %s\n", (a.size() == b.size() && memcmp(a.data(), b.data(), a.size() * 4) == 0) ? "yes" : "NO");
    return 0;
}
