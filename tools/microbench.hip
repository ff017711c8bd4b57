// microbench.hip — VALU issue-rate probes for gfx950 (design data for the warp kernels; not part of the product).
// Each kernel runs ITER iterations of UNROLL independent dependency chains per lane; reports lane-ops/s.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <math.h>
#include "../gyroflow_amd/csrc/gfw_math.h"

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
typedef float f2 __attribute__((ext_vector_type(2)));
constexpr int ITER = 4096;

template <int OP> __global__ __launch_bounds__(256) void k(float *out, float a0, float b0) {
    float v[8]; f2 p[8];
    const float t = (float)threadIdx.x * 1e-7f;
    for (int i = 0; i < 8; ++i) { v[i] = a0 + t + i * 0.001f; p[i] = (f2){v[i], v[i] + 0.5f}; }
    const float b = b0; const f2 pb = {b0, b0 * 1.0001f};
    unsigned long long mask = __ballot(threadIdx.x & 1), m2 = 0;
    for (int it = 0; it < ITER; ++it) {
        #pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (OP == 0) v[i] = __builtin_fmaf(v[i], b, 0.25f);
            if (OP == 1) asm volatile("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(p[i]) : "v"(p[i]), "v"(pb), "v"(pb));
            if (OP == 2) asm volatile("v_mul_f32 %0, %1, %2" : "=v"(v[i]) : "v"(v[i]), "v"(b));
            if (OP == 3) asm volatile("v_add_f32 %0, %1, %2" : "=v"(v[i]) : "v"(v[i]), "v"(b));
            if (OP == 4) asm volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(p[i]) : "v"(p[i]), "v"(pb));
            if (OP == 5) asm volatile("v_pk_add_f32 %0, %1, %2" : "=v"(p[i]) : "v"(p[i]), "v"(pb));
            if (OP == 6) asm volatile("v_rcp_f32 %0, %1" : "=v"(v[i]) : "v"(v[i]));
            if (OP == 7) asm volatile("v_sqrt_f32 %0, %1" : "=v"(v[i]) : "v"(v[i]));
            if (OP == 8) asm volatile("v_rsq_f32 %0, %1" : "=v"(v[i]) : "v"(v[i]));
            if (OP == 9) asm volatile("v_cndmask_b32 %0, %1, %2, vcc" : "=v"(v[i]) : "v"(v[i]), "v"(b));
            if (OP == 10) asm volatile("v_cvt_i32_f32 %0, %1" : "=v"(v[i]) : "v"(v[i]));
            if (OP == 11) v[i] = v[i] / b + 1.0f;            // IEEE division (compiler expansion) + add
            if (OP == 12) v[i] = sqrtf(v[i]) + 2.0f;         // IEEE sqrt
            if (OP == 13) v[i] = atanf(v[i]) + 1.0f;         // ocml atanf
            if (OP == 14) v[i] = gfw_atanf(v[i]) + 1.0f;     // glibc-sequence atanf
            if (OP == 15) asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(v[i]) : "v"(v[i]), "v"(b), "v"(b));
            if (OP == 16) asm volatile("v_rndne_f32 %0, %1" : "=v"(v[i]) : "v"(v[i]));
            if (OP == 17) asm volatile("v_and_b32 %0, %1, %2" : "=v"(v[i]) : "v"(v[i]), "v"(b));
            if (OP == 18) asm volatile("v_mad_u32_u24 %0, %1, %2, %3" : "=v"(v[i]) : "v"(v[i]), "v"(b), "v"(b));
            if (OP == 19) asm volatile("v_max_f32 %0, %1, %2" : "=v"(v[i]) : "v"(v[i]), "v"(b));
            if (OP == 20) asm volatile("v_cmp_lt_f32 vcc, %0, %1" :: "v"(v[i]), "v"(b) : "vcc");
            if (OP == 21) asm volatile("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(v[i]) : "v"(v[i]), "v"(b), "s"(mask));
            if (OP == 22) asm volatile("v_cmp_lt_f32 vcc, %1, %2\n\tv_cndmask_b32 %0, %1, %2, vcc" : "=v"(v[i]) : "v"(v[i]), "v"(b) : "vcc");
            if (OP == 23) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(v[i]) : "v"(b), "v"(b));
            if (OP == 24) asm volatile("v_fma_f32 %0, -%1, %2, %3" : "=v"(v[i]) : "v"(v[i]), "v"(b), "v"(b));
            if (OP == 25) asm volatile("v_min_f32 %0, %1, %2" : "=v"(v[i]) : "v"(v[i]), "v"(b));
            if (OP == 26) asm volatile("v_sub_f32 %0, %1, %2" : "=v"(v[i]) : "v"(v[i]), "v"(b));
            if (OP == 27) asm volatile("v_cvt_f32_u32 %0, %1" : "=v"(v[i]) : "v"(v[i]));
            if (OP == 28) asm volatile("v_bfi_b32 %0, %1, %2, %3" : "=v"(v[i]) : "v"(v[i]), "v"(b), "v"(b));
            if (OP == 29) asm volatile("v_lshlrev_b32 %0, 1, %1" : "=v"(v[i]) : "v"(v[i]));
            if (OP == 30) asm volatile("v_cmp_lt_f32 %0, %1, %2" : "=s"(m2) : "v"(v[i]), "v"(b));
            if (OP == 31) asm volatile("v_mov_b32 %0, %1" : "=v"(v[i]) : "v"(v[i]));
            if (OP == 32) asm volatile("v_mul_f32 %0, %1, %2" : "=v"(v[i]) : "s"(b), "v"(v[i]));
            if (OP == 33) asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(v[i]) : "v"(v[i]), "s"(b), "v"(v[i]));
            if (OP == 34) asm volatile("v_add_u32 %0, %1, %2" : "=v"(v[i]) : "v"(v[i]), "v"(b));
            if (OP == 36) asm volatile("v_mul_f32 %0, 2.0, %1" : "=v"(v[i]) : "v"(v[i]));
            if (OP == 37) asm volatile("v_mul_f32 %0, 0x3f7fbe77, %1" : "=v"(v[i]) : "v"(v[i]));
            if (OP == 38) asm volatile("v_add_f32 %0, 0x3a83126f, %1" : "=v"(v[i]) : "v"(v[i]));
            if (OP == 39) asm volatile("v_add_f32 %0, %1, %2" : "=v"(v[i]) : "s"(b), "v"(v[i]));
            if (OP == 40) asm volatile("v_fmac_f32 %0, 0x3f7fbe77, %1" : "+v"(v[i]) : "v"(b));
            if (OP == 41) asm volatile("v_fmaak_f32 %0, %1, %2, 0x3e800000" : "=v"(v[i]) : "v"(v[i]), "v"(b));
            if (OP == 35) asm volatile("v_cvt_f32_u32_sdwa %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1" : "=v"(v[i]) : "v"(v[i]));
        }
    }
    float s = (float)(m2 & 1); for (int i = 0; i < 8; ++i) s += v[i] + p[i].x + p[i].y;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int OP> double run(const char *name, float *d, int lanes_per_op, float a0, float b0, int blocks_per_cu) {
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    const int grid = 256 * blocks_per_cu;
    hipLaunchKernelGGL(k<OP>, dim3(grid), dim3(256), 0, 0, d, a0, b0);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(k<OP>, dim3(grid), dim3(256), 0, 0, d, a0, b0);
    CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1)); ms /= 5;
    const double ops = (double)grid * 256 * ITER * 8;           // lane-instructions
    const double rate = ops / (ms * 1e-3);
    printf("%-28s %8.3f ms  %7.2f T lane-instr/s  (%6.2f G wave-instr/s, x%d elems = %7.2f T elem-ops/s)\n", name, ms, rate / 1e12, rate / 64 / 1e9, lanes_per_op, rate * lanes_per_op / 1e12);
    return rate;
}

int main() {
    float *d; CHECK(hipMalloc(&d, 256 * 16 * 256 * sizeof(float)));
    hipDeviceProp_t pr; CHECK(hipGetDeviceProperties(&pr, 0));
    printf("device: %s %s CUs=%d clock=%d MHz\n", pr.name, pr.gcnArchName, pr.multiProcessorCount, pr.clockRate / 1000);
    for (int bpc : {4, 8}) {
        printf("--- %d blocks (x256 threads) per CU\n", bpc);
        run<0>("v_fma_f32 (compiler)", d, 1, 1.0f, 0.999f, bpc);
        run<15>("v_fma_f32 (asm)", d, 1, 1.0f, 0.999f, bpc);
        run<1>("v_pk_fma_f32", d, 2, 1.0f, 0.999f, bpc);
        run<2>("v_mul_f32", d, 1, 1.0f, 0.999f, bpc);
        run<3>("v_add_f32", d, 1, 1.0f, 0.001f, bpc);
        run<4>("v_pk_mul_f32", d, 2, 1.0f, 0.999f, bpc);
        run<5>("v_pk_add_f32", d, 2, 1.0f, 0.001f, bpc);
        run<6>("v_rcp_f32", d, 1, 1.5f, 1.0f, bpc);
        run<7>("v_sqrt_f32", d, 1, 1.5f, 1.0f, bpc);
        run<8>("v_rsq_f32", d, 1, 1.5f, 1.0f, bpc);
        run<9>("v_cndmask_b32", d, 1, 1.5f, 1.0f, bpc);
        run<10>("v_cvt_i32_f32", d, 1, 1.5f, 1.0f, bpc);
        run<16>("v_rndne_f32", d, 1, 1.5f, 1.0f, bpc);
        run<17>("v_and_b32", d, 1, 1.5f, 1.0f, bpc);
        run<18>("v_mad_u32_u24", d, 1, 1.5f, 1.0f, bpc);
        run<19>("v_max_f32", d, 1, 1.5f, 1.0f, bpc);
        run<20>("v_cmp_lt_f32", d, 1, 1.5f, 1.0f, bpc);
        run<21>("v_cndmask_b32_e64 (sgpr mask)", d, 1, 1.5f, 1.0f, bpc);
        run<22>("v_cmp + v_cndmask (vcc) pair", d, 1, 1.5f, 1.0f, bpc);
        run<30>("v_cmp_lt_f32_e64 -> sgpr", d, 1, 1.5f, 1.0f, bpc);
        run<23>("v_fmac_f32 (asm)", d, 1, 1.0f, 0.001f, bpc);
        run<24>("v_fma_f32 neg mod (asm)", d, 1, 1.0f, 0.999f, bpc);
        run<33>("v_fma_f32 sgpr src (asm)", d, 1, 1.0f, 0.5f, bpc);
        run<32>("v_mul_f32 sgpr src", d, 1, 1.0f, 0.999f, bpc);
        run<36>("v_mul_f32 inline const", d, 1, 1.0f, 0.999f, bpc);
        run<37>("v_mul_f32 literal", d, 1, 1.0f, 0.999f, bpc);
        run<38>("v_add_f32 literal", d, 1, 1.0f, 0.999f, bpc);
        run<39>("v_add_f32 sgpr src", d, 1, 1.0f, 0.001f, bpc);
        run<40>("v_fmac_f32 literal", d, 1, 1.0f, 0.001f, bpc);
        run<41>("v_fmaak_f32", d, 1, 1.0f, 0.999f, bpc);
        run<25>("v_min_f32", d, 1, 1.5f, 1.0f, bpc);
        run<26>("v_sub_f32", d, 1, 1.5f, 0.001f, bpc);
        run<27>("v_cvt_f32_u32", d, 1, 1.5f, 1.0f, bpc);
        run<35>("v_cvt_f32_u32_sdwa", d, 1, 1.5f, 1.0f, bpc);
        run<28>("v_bfi_b32", d, 1, 1.5f, 1.0f, bpc);
        run<29>("v_lshlrev_b32", d, 1, 1.5f, 1.0f, bpc);
        run<34>("v_add_u32", d, 1, 1.5f, 1.0f, bpc);
        run<31>("v_mov_b32", d, 1, 1.5f, 1.0f, bpc);
        run<11>("IEEE a/b + add", d, 1, 1.5f, 1.0001f, bpc);
        run<12>("IEEE sqrtf + add", d, 1, 1.5f, 1.0f, bpc);
        run<13>("ocml atanf + add", d, 1, 0.7f, 1.0f, bpc);
        run<14>("gfw_atanf + add", d, 1, 0.7f, 1.0f, bpc);
    }
    return 0;
}
