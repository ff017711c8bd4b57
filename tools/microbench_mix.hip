// microbench_mix.hip — what a VALU instruction costs INSIDE a mix (design data for the warp kernels; not part of the product).
// tools/microbench.hip measures each opcode alone on 8 independent chains per lane (fast class 2.3 cycles, the rest 4.3, packed 4.6).  The warp kernel
// runs ~224 instructions per pixel at 4.4 SIMD-cycles per instruction although a third of them are fast-class: this probe measures sequences —
// alternating classes, dependent chains, packed forms beside scalar ones, SALU beside VALU — at 8 waves per SIMD, and prints SIMD-cycles per instruction.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
typedef float f2 __attribute__((ext_vector_type(2)));
constexpr int ITER = 2048;

// NCH independent chains per lane; PAT selects the instruction sequence applied to each chain per iteration; returns instructions per chain-iteration
template <int PAT, int NCH> __global__ __launch_bounds__(256) void k(float *out, float a0, float b0) {
    float v[8]; f2 p[8]; int q[8];
    const float t = (float)threadIdx.x * 1e-7f;
    for (int i = 0; i < 8; ++i) { v[i] = a0 + t + i * 0.001f; p[i] = (f2){v[i], v[i] + 0.5f}; q[i] = threadIdx.x + i; }
    const float b = b0; const f2 pb = {b0, b0 * 1.0001f};
    int sacc = 0;
    __shared__ float lds[256]; lds[threadIdx.x] = t; __syncthreads();
    for (int it = 0; it < ITER; ++it) {
        #pragma unroll
        for (int i = 0; i < NCH; ++i) {
            if (PAT == 0) { asm volatile("v_mul_f32 %0, %1, %2" : "=v"(v[i]) : "v"(v[i]), "v"(b)); asm volatile("v_add_f32 %0, %1, %2" : "=v"(v[i]) : "v"(v[i]), "v"(b)); }
            if (PAT == 1) { asm volatile("v_mul_f32 %0, %1, %2" : "=v"(v[i]) : "v"(v[i]), "v"(b)); asm volatile("v_max_f32 %0, %1, %2" : "=v"(v[i]) : "v"(v[i]), "v"(b)); }
            if (PAT == 2) { asm volatile("v_max_f32 %0, %1, %2" : "=v"(v[i]) : "v"(v[i]), "v"(b)); asm volatile("v_min_f32 %0, %1, %2" : "=v"(v[i]) : "v"(v[i]), "v"(b)); }
            if (PAT == 3) { asm volatile("v_mul_f32 %0, %1, %2" : "=v"(v[i]) : "v"(v[i]), "v"(b)); asm volatile("v_add_f32 %0, %1, %2" : "=v"(v[i]) : "v"(v[i]), "v"(b));
                            asm volatile("v_mul_f32 %0, %1, %2" : "=v"(v[i]) : "v"(v[i]), "v"(b)); asm volatile("v_max_f32 %0, %1, %2" : "=v"(v[i]) : "v"(v[i]), "v"(b)); }
            if (PAT == 4) { asm volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(p[i]) : "v"(p[i]), "v"(pb)); asm volatile("v_pk_add_f32 %0, %1, %2" : "=v"(p[i]) : "v"(p[i]), "v"(pb)); }
            if (PAT == 5) { asm volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(p[i]) : "v"(p[i]), "v"(pb)); asm volatile("v_mul_f32 %0, %1, %2" : "=v"(v[i]) : "v"(v[i]), "v"(b)); }
            if (PAT == 6) { asm volatile("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(p[i]) : "v"(p[i]), "v"(pb), "v"(pb)); asm volatile("v_max_f32 %0, %1, %2" : "=v"(v[i]) : "v"(v[i]), "v"(b)); }
            if (PAT == 7) { asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(v[i]) : "v"(v[i]), "v"(b), "v"(v[i])); asm volatile("v_fma_f32 %0, -%1, %2, %3" : "=v"(v[i]) : "v"(v[i]), "v"(b), "v"(b)); }
            if (PAT == 8) { asm volatile("v_mul_f32 %0, %1, %2" : "=v"(v[i]) : "v"(v[i]), "v"(b)); asm volatile("s_add_u32 %0, %0, 1" : "+s"(sacc) :: "scc"); asm volatile("v_add_f32 %0, %1, %2" : "=v"(v[i]) : "v"(v[i]), "v"(b)); asm volatile("s_add_u32 %0, %0, 3" : "+s"(sacc) :: "scc"); }
            if (PAT == 9) { asm volatile("v_max_f32 %0, %1, %2" : "=v"(v[i]) : "v"(v[i]), "v"(b)); asm volatile("s_add_u32 %0, %0, 1" : "+s"(sacc) :: "scc"); asm volatile("v_min_f32 %0, %1, %2" : "=v"(v[i]) : "v"(v[i]), "v"(b)); asm volatile("s_add_u32 %0, %0, 3" : "+s"(sacc) :: "scc"); }
            if (PAT == 10) { asm volatile("v_mul_f32 %0, %1, %2" : "=v"(v[i]) : "v"(v[i]), "v"(b)); asm volatile("v_cvt_i32_f32 %0, %1" : "=v"(q[i]) : "v"(v[i])); asm volatile("v_cvt_f32_i32 %0, %1" : "=v"(v[i]) : "v"(q[i])); }
            if (PAT == 11) { asm volatile("v_mul_f32 %0, %1, %2" : "=v"(v[i]) : "v"(v[i]), "v"(b)); asm volatile("v_cmp_lt_f32 vcc, %1, %2\n\tv_cndmask_b32 %0, %1, %2, vcc" : "=v"(v[i]) : "v"(v[i]), "v"(b) : "vcc"); }
            if (PAT == 12) { asm volatile("v_mul_f32 %0, %1, %2" : "=v"(v[i]) : "v"(v[i]), "v"(b)); asm volatile("v_rcp_f32 %0, %1" : "=v"(v[i]) : "v"(v[i])); }
            if (PAT == 13) { asm volatile("v_mul_f32 %0, 0x3f7fbe77, %1" : "=v"(v[i]) : "v"(v[i])); asm volatile("v_add_f32 %0, 0x3a83126f, %1" : "=v"(v[i]) : "v"(v[i])); }
            if (PAT == 14) { asm volatile("v_mul_f32 %0, %1, %2" : "=v"(v[i]) : "v"(v[i]), "v"(b)); asm volatile("v_and_b32 %0, 0x7fffffff, %1" : "=v"(v[i]) : "v"(v[i])); asm volatile("v_add_u32 %0, 1, %1" : "=v"(v[i]) : "v"(v[i])); }
            if (PAT == 15) { asm volatile("v_mul_f32 %0, %1, %2" : "=v"(v[i]) : "v"(v[i]), "v"(b)); asm volatile("v_lshlrev_b32 %0, 1, %1" : "=v"(v[i]) : "v"(v[i])); }
            if (PAT == 16) { asm volatile("v_mov_b32 %0, %1" : "=v"(v[i]) : "v"(v[(i + 1) & (NCH - 1)])); asm volatile("v_mul_f32 %0, %1, %2" : "=v"(v[i]) : "v"(v[i]), "v"(b)); }

            if (PAT == 18) { if (v[i] > -1e30f) { asm volatile("v_mul_f32 %0, %1, %2" : "=v"(v[i]) : "v"(v[i]), "v"(b)); } asm volatile("v_add_f32 %0, %1, %2" : "=v"(v[i]) : "v"(v[i]), "v"(b)); }
            if (PAT == 19) { asm volatile("v_mul_f32 %0, %1, %2" : "=v"(v[i]) : "v"(v[i]), "s"(b)); asm volatile("v_add_f32 %0, %1, %2" : "=v"(v[i]) : "v"(v[i]), "s"(b)); }
            if (PAT == 20) { asm volatile("v_mul_f32 %0, %1, %2" : "=v"(v[i]) : "v"(v[i]), "v"(b)); v[i] += lds[(threadIdx.x + it) & 255]; }
            if (PAT == 21) { asm volatile("v_mul_f32 %0, %1, %2" : "=v"(v[i]) : "v"(v[i]), "v"(b)); asm volatile("s_nop 0"); asm volatile("v_add_f32 %0, %1, %2" : "=v"(v[i]) : "v"(v[i]), "v"(b)); asm volatile("s_nop 0"); }
            if (PAT == 22) { asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(v[i]) : "v"(b), "v"(b)); asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(v[i]) : "v"(v[i]), "v"(b), "v"(p[i].x)); }
            if (PAT == 23) { asm volatile("v_mul_f32 %0, %1, %2" : "=v"(v[i]) : "v"(v[i]), "v"(b)); asm volatile("v_max_f32 %0, %1, %2" : "=v"(v[i]) : "v"(v[i]), "v"(b)); asm volatile("v_min_f32 %0, %1, %2" : "=v"(v[i]) : "v"(v[i]), "v"(b)); }
            if (PAT == 24) { asm volatile("v_mul_f32 %0, %1, %2" : "=v"(v[i]) : "v"(v[i]), "v"(b)); asm volatile("v_add_f32 %0, %1, %2" : "=v"(v[i]) : "v"(v[i]), "v"(b)); asm volatile("v_max_f32 %0, %1, %2" : "=v"(v[i]) : "v"(v[i]), "v"(b)); asm volatile("v_min_f32 %0, %1, %2" : "=v"(v[i]) : "v"(v[i]), "v"(b)); asm volatile("v_cvt_i32_f32 %0, %1" : "=v"(q[i]) : "v"(v[i])); }
            if (PAT == 17) { asm volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(p[i]) : "v"(p[i]), "v"(pb)); asm volatile("v_pk_add_f32 %0, %1, %2" : "=v"(p[i]) : "v"(p[i]), "v"(pb)); asm volatile("v_max_f32 %0, %1, %2" : "=v"(v[i]) : "v"(v[i]), "v"(b)); }
        }
    }
    float s = (float)sacc; for (int i = 0; i < 8; ++i) s += v[i] + p[i].x + p[i].y + (float)q[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

static int g_mhz = 2400;
template <int PAT, int NCH> void run(const char *name, float *d, int instr_per_chain_iter, int bpc) {
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    const int grid = 256 * bpc;
    hipLaunchKernelGGL((k<PAT, NCH>), dim3(grid), dim3(256), 0, 0, d, 1.0f, 0.999f);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    for (int r = 0; r < 5; ++r) hipLaunchKernelGGL((k<PAT, NCH>), dim3(grid), dim3(256), 0, 0, d, 1.0f, 0.999f);
    CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1)); ms /= 5;
    const double wave_instr = (double)grid * 4 * ITER * NCH * instr_per_chain_iter;        // VALU wave-instructions (SALU not counted)
    const double simd_cycles = ms * 1e-3 * g_mhz * 1e6 * 1024.0;
    printf("%-64s chains=%d wg/CU=%d  %7.3f ms   %.2f SIMD-cycles per VALU instruction (at %d MHz nominal)\n", name, NCH, bpc, ms, simd_cycles / wave_instr, g_mhz);
}

int main() {
    float *d; CHECK(hipMalloc(&d, 256 * 16 * 256 * sizeof(float)));
    hipDeviceProp_t pr; CHECK(hipGetDeviceProperties(&pr, 0));
    g_mhz = pr.clockRate / 1000;
    printf("device: %s %s CUs=%d clock=%d MHz\n", pr.name, pr.gcnArchName, pr.multiProcessorCount, g_mhz);
    for (int bpc : {8}) {
#define BOTH(PAT, name, n) run<PAT, 8>(name, d, n, bpc); run<PAT, 1>(name, d, n, bpc);
        BOTH(0, "mul, add (fast, fast)", 2)
        BOTH(13, "mul literal, add literal (fast, fast)", 2)
        BOTH(1, "mul, max (fast, slow)", 2)
        BOTH(2, "max, min (slow, slow)", 2)
        BOTH(3, "mul, add, mul, max (3 fast, 1 slow)", 4)
        BOTH(4, "pk_mul, pk_add", 2)
        BOTH(5, "pk_mul, mul (packed, fast)", 2)
        BOTH(6, "pk_fma, max (packed, slow)", 2)
        BOTH(17, "pk_mul, pk_add, max", 3)
        BOTH(7, "fma 3-operand, fma neg (slow, slow)", 2)
        BOTH(8, "mul, s_add, add, s_add (fast + SALU)", 2)
        BOTH(9, "max, s_add, min, s_add (slow + SALU)", 2)
        BOTH(10, "mul, cvt_i32_f32, cvt_f32_i32 (fast, slow, slow)", 3)
        BOTH(11, "mul, cmp, cndmask", 3)
        BOTH(12, "mul, rcp (fast, trans)", 2)
        BOTH(14, "mul, and, add_u32 (fast x3)", 3)
        BOTH(15, "mul, lshl (fast, slow)", 2)
        BOTH(16, "mov (cross-chain), mul", 2)
        BOTH(18, "if (always true, per lane) { mul } add", 2)
        BOTH(19, "mul sgpr, add sgpr", 2)
        BOTH(20, "mul, ds_read + add", 2)
        BOTH(21, "mul, s_nop, add, s_nop", 2)
        BOTH(22, "fmac, fma 3 distinct operands", 2)
        BOTH(23, "mul, max, min (1 fast, 2 slow)", 3)
        BOTH(24, "mul, add, max, min, cvt (2 fast, 3 slow)", 5)
    }
    return 0;
}
