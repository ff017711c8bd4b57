// microbench_cycles.hip — true shader cycles per wave-instruction, by class, at 1 / 2 / 4 / 8 waves per SIMD (design data for the warp kernels; not part of the product).
// tools/microbench_mix.hip divided wall time by a NOMINAL clock at one occupancy; this probe reads s_memtime (tick = shader cycle, MI355X_MICROARCH.md) around the
// measured loop inside every wave, records which SIMD the wave ran on (HW_REG_HW_ID / HW_REG_XCC_ID), and reports, per SIMD, issued wave-instructions / busy cycles:
//   cycles per wave-instruction on one SIMD = (last end - first start over the SIMD's waves) / (sum of their instructions)      [only SIMDs that held exactly W waves count]
// and the single wave's own view (its delta / its instructions) beside it.  Build: hipcc --offload-arch=gfx950 -O2 tools/microbench_cycles.hip -o tools/microbench_cycles
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <map>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
constexpr int ITER = 512, NCH = 8;

struct Rec { unsigned long long t0, t1; unsigned hw, xcc; };

template <int PAT> __global__ __launch_bounds__(256) void k(Rec *rec, float *out, float a0, float b0, const unsigned *mem) {
    float v[NCH], w[NCH]; unsigned q[NCH]; unsigned long long m64 = ~0ull;
    const float t = (float)threadIdx.x * 1e-7f;
    for (int i = 0; i < NCH; ++i) { v[i] = a0 + t + i * 0.001f; w[i] = v[i] * 0.5f; q[i] = threadIdx.x * 2654435761u + i; }
    const float b = b0;
    float sb; asm volatile("v_readfirstlane_b32 %0, %1" : "=s"(sb) : "v"(b));
    int sacc = 0;
    __shared__ float lds[512]; lds[threadIdx.x] = t; lds[threadIdx.x + 256] = t; __syncthreads();
    unsigned long long t0, t1;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0) :: "memory");
    #pragma unroll 1
    for (int it = 0; it < ITER; ++it) {
      // each pattern is a sequence of STAGES; a stage issues ONE instruction on each of the NCH independent chains, so neighbouring instructions never depend on each other
      #pragma unroll
      for (int st = 0; st < 4; ++st) {
        #pragma unroll
        for (int i = 0; i < NCH; ++i) {
            if (PAT == 0) { if (st == 0) asm volatile("v_mul_f32 %0, %1, %2" : "=v"(v[i]) : "v"(v[i]), "v"(b)); if (st == 1) asm volatile("v_add_f32 %0, %1, %2" : "=v"(v[i]) : "v"(v[i]), "v"(b)); }
            if (PAT == 1) { if (st < 2) asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(v[i]) : "v"(v[i]), "v"(b), "v"(b)); }
            if (PAT == 27) { if (st < 2) asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(v[i]) : "v"(v[i]), "v"(b), "v"(v[(i + 3) & (NCH - 1)])); }
            if (PAT == 28) { if (st < 2) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(v[i]) : "v"(b), "v"(v[(i + 3) & (NCH - 1)])); }
            if (PAT == 2) { if (st == 0) asm volatile("v_max_f32 %0, %1, %2" : "=v"(v[i]) : "v"(v[i]), "v"(b)); if (st == 1) asm volatile("v_min_f32 %0, %1, %2" : "=v"(v[i]) : "v"(v[i]), "v"(b)); }
            if (PAT == 3) { if (st == 0) asm volatile("v_cvt_f32_u32 %0, %1" : "=v"(v[i]) : "v"(q[i])); if (st == 1) asm volatile("v_cvt_u32_f32 %0, %1" : "=v"(q[i]) : "v"(v[i])); }
            if (PAT == 4) { if (st == 0) asm volatile("v_cvt_f32_u32_sdwa %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0" : "=v"(v[i]) : "v"(q[i])); if (st == 1) asm volatile("v_cvt_f32_u32_sdwa %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1" : "=v"(v[i]) : "v"(q[i])); }
            if (PAT == 5) { if (st == 0) asm volatile("v_mul_f32 %0, %1, %2" : "=v"(v[i]) : "s"(sb), "v"(v[i])); if (st == 1) asm volatile("v_add_f32 %0, %1, %2" : "=v"(v[i]) : "s"(sb), "v"(v[i])); }
            if (PAT == 6) { if (st == 0) asm volatile("v_mul_f32 %0, %1, %2" : "=v"(v[i]) : "v"(v[i]), "v"(b)); if (st == 1) asm volatile("s_add_u32 %0, %0, 1" : "+s"(sacc) :: "scc"); }
            if (PAT == 26) { if (st == 0) asm volatile("v_mul_f32 %0, %1, %2" : "=v"(v[i]) : "v"(v[i]), "v"(b)); if (st == 1) asm volatile("s_add_u32 %0, %0, 1" : "+s"(sacc) :: "scc"); if (st == 2) asm volatile("s_and_b32 %0, %0, 0xffff" : "+s"(sacc) :: "scc"); }
            if (PAT == 29) { if (st == 0) asm volatile("v_mul_f32 %0, %1, %2" : "=v"(v[i]) : "v"(v[i]), "v"(b)); if (st == 1) asm volatile("s_and_b64 %0, %0, exec" : "+s"(m64) :: "scc"); }
            if (PAT == 16) { if (st == 0) asm volatile("v_mul_f32 %0, %1, %2" : "=v"(v[i]) : "v"(v[i]), "v"(b)); if (st == 1) asm volatile("s_nop 0"); }
            if (PAT == 7) { if (st == 0) asm volatile("v_cmp_lt_f32 vcc, %1, %2\n\tv_cndmask_b32 %0, %1, %2, vcc" : "=v"(v[i]) : "v"(v[i]), "v"(b) : "vcc"); }
            if (PAT == 8) { if (st == 0) { unsigned long long m; asm volatile("v_cmp_lt_f32_e64 %0, %1, %2" : "=s"(m) : "v"(v[i]), "v"(b)); asm volatile("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(v[i]) : "v"(v[i]), "v"(b), "s"(m)); } }
            if (PAT == 30) { if (st == 0) { unsigned long long m; asm volatile("v_cmp_lt_f32_e64 %0, %1, %2" : "=s"(m) : "v"(v[i]), "v"(b)); asm volatile("s_and_b64 %0, %0, %1" : "+s"(m64) : "s"(m) : "scc"); } }
            if (PAT == 9) { if (st == 0) asm volatile("v_and_b32 %0, 0x7fffffff, %1" : "=v"(q[i]) : "v"(q[i])); if (st == 1) asm volatile("v_add_u32 %0, 1, %1" : "=v"(q[i]) : "v"(q[i])); }
            if (PAT == 10) { if (st == 0) asm volatile("v_lshlrev_b32 %0, 1, %1" : "=v"(q[i]) : "v"(q[i])); if (st == 1) asm volatile("v_lshrrev_b32 %0, 1, %1" : "=v"(q[i]) : "v"(q[i])); }
            if (PAT == 31) { if (st == 0) asm volatile("v_ashrrev_i32 %0, 5, %1" : "=v"(q[i]) : "v"(q[i])); if (st == 1) asm volatile("v_bfe_u32 %0, %1, 0, 5" : "=v"(q[i]) : "v"(q[i])); }
            if (PAT == 11) { if (st == 0) asm volatile("v_mad_u32_u24 %0, %1, %2, %3" : "=v"(q[i]) : "v"(q[i]), "v"(q[i]), "v"(q[i])); if (st == 1) asm volatile("v_mul_u32_u24 %0, %1, %2" : "=v"(q[i]) : "v"(q[i]), "v"(q[i])); }
            if (PAT == 12) { if (st == 0) asm volatile("v_rndne_f32 %0, %1" : "=v"(v[i]) : "v"(v[i])); if (st == 1) asm volatile("v_med3_f32 %0, %1, %2, %3" : "=v"(v[i]) : "v"(v[i]), "v"(b), "v"(b)); }
            if (PAT == 13) { if (st == 0) asm volatile("v_rcp_f32 %0, %1" : "=v"(v[i]) : "v"(v[i])); if (st == 1) asm volatile("v_sqrt_f32 %0, %1" : "=v"(v[i]) : "v"(v[i])); }
            if (PAT == 14) { if (i < NCH / 2) { if (st == 0) asm volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(*(double *)&v[2 * i]) : "v"(*(double *)&v[2 * i]), "v"(*(double *)&v[2 * i])); if (st == 1) asm volatile("v_pk_add_f32 %0, %1, %2" : "=v"(*(double *)&v[2 * i]) : "v"(*(double *)&v[2 * i]), "v"(*(double *)&v[2 * i])); } }
            if (PAT == 15) { if (st == 0) asm volatile("v_mov_b32 %0, %1" : "=v"(v[i]) : "v"(v[(i + 1) & (NCH - 1)])); if (st == 1) asm volatile("v_mov_b32_dpp %0, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "=v"(v[i]) : "v"(v[i]), "0"(v[i])); }
            if (PAT == 17) { if (st == 0) asm volatile("v_mul_f32 %0, %1, %2" : "=v"(v[i]) : "v"(v[i]), "v"(b)); if (st == 1) v[i] += lds[(threadIdx.x + it + i) & 511]; }
            if (PAT == 18) { if (st == 0) asm volatile("v_mul_f32 %0, 0x3f7fbe77, %1" : "=v"(v[i]) : "v"(v[i])); if (st == 1) asm volatile("v_add_f32 %0, 0x3a83126f, %1" : "=v"(v[i]) : "v"(v[i])); }
            if (PAT == 19) { if (st == 0) asm volatile("v_cvt_f32_ubyte0 %0, %1" : "=v"(v[i]) : "v"(q[i])); if (st == 1) asm volatile("v_cvt_f32_ubyte2 %0, %1" : "=v"(v[i]) : "v"(q[i])); }
            // a Lanczos4 tap as the kernel issues it: convert (SDWA), multiply by the weight, add to the row sum — three independent instructions per chain and stage
            if (PAT == 20) { if (st == 0) asm volatile("v_cvt_f32_u32_sdwa %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0" : "=v"(w[i]) : "v"(q[i])); if (st == 1) asm volatile("v_mul_f32 %0, %1, %2" : "=v"(w[i]) : "v"(w[i]), "v"(b)); if (st == 2) asm volatile("v_add_f32 %0, %1, %2" : "=v"(v[i]) : "v"(v[i]), "v"(w[i])); }
            // ... and with the conversion done by the memory pipeline (what a typed load would leave for the VALU): multiply, add
            if (PAT == 32) { if (st == 0) asm volatile("v_mul_f32 %0, %1, %2" : "=v"(w[i]) : "v"(w[i]), "v"(b)); if (st == 1) asm volatile("v_add_f32 %0, %1, %2" : "=v"(v[i]) : "v"(v[i]), "v"(w[i])); }
            if (PAT == 21) { if (st == 0) asm volatile("v_alignbit_b32 %0, %1, %2, 16" : "=v"(q[i]) : "v"(q[i]), "v"(q[(i + 1) & (NCH - 1)])); if (st == 1) asm volatile("v_perm_b32 %0, %1, %2, %3" : "=v"(q[i]) : "v"(q[i]), "v"(q[i]), "v"(q[(i + 1) & (NCH - 1)])); }
            if (PAT == 22) { if (st == 0) asm volatile("v_cvt_i32_f32 %0, %1" : "=v"(q[i]) : "v"(v[i])); if (st == 1) asm volatile("v_cvt_f32_i32 %0, %1" : "=v"(v[i]) : "v"(q[i])); }
            if (PAT == 23) { if (st == 0) asm volatile("v_readlane_b32 %0, %1, 3" : "=s"(sacc) : "v"(q[i])); if (st == 1) asm volatile("v_add_u32 %0, %1, %2" : "=v"(q[i]) : "s"(sacc), "v"(q[i])); }
            // two full-rate and two half-rate instructions per chain: do the classes overlap?
            if (PAT == 24) { if (st == 0) asm volatile("v_mul_f32 %0, %1, %2" : "=v"(v[i]) : "v"(v[i]), "v"(b)); if (st == 1) asm volatile("v_max_f32 %0, %1, %2" : "=v"(v[i]) : "v"(v[i]), "v"(b)); if (st == 2) asm volatile("v_add_f32 %0, %1, %2" : "=v"(v[i]) : "v"(v[i]), "v"(b)); if (st == 3) asm volatile("v_cvt_i32_f32 %0, %1" : "=v"(q[i]) : "v"(v[i])); }
            if (PAT == 25) { if (st == 0) asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(v[i]) : "v"(v[i]), "s"(sb), "v"(b)); if (st == 1) asm volatile("v_fma_f32 %0, -%1, %2, %3" : "=v"(v[i]) : "v"(v[i]), "v"(b), "v"(v[i])); }
        }
      }
    }
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1) :: "memory");
    unsigned hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    float s = (float)sacc + (float)(m64 & 0xff); for (int i = 0; i < NCH; ++i) s += v[i] + w[i] + (float)q[i];
    const unsigned gid = blockIdx.x * blockDim.x + threadIdx.x;
    out[gid] = s + (mem ? (float)mem[gid & 255] : 0.0f);
    if ((threadIdx.x & 63) == 0) rec[gid >> 6] = Rec{t0, t1, hw, xcc};
}

template <int PAT> void run(const char *name, int instr_per_chain_iter, Rec *d_rec, float *d_out) {
    printf("%-58s", name);
    for (int W : {1, 2, 4, 8}) {
        const int grid = 256 * W;
        std::vector<Rec> h((size_t)grid * 4);
        hipLaunchKernelGGL((k<PAT>), dim3(grid), dim3(256), 0, 0, d_rec, d_out, 1.0f, 0.999f, (const unsigned *)nullptr);     // warm
        CHECK(hipDeviceSynchronize());
        hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL((k<PAT>), dim3(grid), dim3(256), 0, 0, d_rec, d_out, 1.0f, 0.999f, (const unsigned *)nullptr);
        CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
        CHECK(hipMemcpy(h.data(), d_rec, h.size() * sizeof(Rec), hipMemcpyDeviceToHost));
        const double n_instr = (double)ITER * NCH * instr_per_chain_iter;
        // per SIMD: the waves that ran there
        std::map<unsigned long long, std::vector<const Rec *>> simd;
        for (const Rec &r : h) simd[((unsigned long long)(r.xcc & 0xf) << 32) | (r.hw & 0xfff0u & ~0xc0u)].push_back(&r);   // key: xcc | se, sh, cu, simd (wave and pipe ids masked out)
        std::vector<double> per_simd, own;
        for (auto &kv : simd) {
            if ((int)kv.second.size() != W) continue;
            unsigned long long a = ~0ull, b = 0;
            for (const Rec *r : kv.second) { a = std::min(a, r->t0); b = std::max(b, r->t1); own.push_back((double)(r->t1 - r->t0) / n_instr); }
            per_simd.push_back((double)(b - a) / (n_instr * W));
        }
        std::sort(per_simd.begin(), per_simd.end()); std::sort(own.begin(), own.end());
        if (per_simd.empty()) { printf("  W=%d: (no SIMD held exactly %d waves; %zu SIMDs seen)", W, W, simd.size()); continue; }
        printf("  W=%d: %5.2f (own %5.2f, %3zu SIMDs)", W, per_simd[per_simd.size() / 2], own[own.size() / 2], per_simd.size());
    }
    printf("\n");
}

int main() {
    Rec *d_rec; float *d_out;
    CHECK(hipMalloc(&d_rec, 256 * 8 * 4 * sizeof(Rec))); CHECK(hipMalloc(&d_out, 256 * 8 * 256 * sizeof(float)));
    hipDeviceProp_t pr; CHECK(hipGetDeviceProperties(&pr, 0));
    printf("device: %s %s CUs=%d nominal clock=%d MHz\n", pr.name, pr.gcnArchName, pr.multiProcessorCount, pr.clockRate / 1000);
    printf("shader cycles (s_memtime) per wave-instruction on ONE SIMD at W waves per SIMD, median over the SIMDs that held exactly W waves; 'own' = one wave's delta / its own instructions\n");
    run<0>("v_mul_f32, v_add_f32", 2, d_rec, d_out);
    run<18>("v_mul_f32 literal, v_add_f32 literal", 2, d_rec, d_out);
    run<1>("v_fma_f32 v,b,b (2 distinct VGPRs)", 2, d_rec, d_out);
    run<27>("v_fma_f32 v,b,c (3 distinct VGPRs)", 2, d_rec, d_out);
    run<28>("v_fmac_f32", 2, d_rec, d_out);
    run<25>("v_fma_f32 v,s,b ; v_fma_f32 -v,b,v", 2, d_rec, d_out);
    run<2>("v_max_f32, v_min_f32", 2, d_rec, d_out);
    run<12>("v_rndne_f32, v_med3_f32", 2, d_rec, d_out);
    run<3>("v_cvt_f32_u32, v_cvt_u32_f32", 2, d_rec, d_out);
    run<22>("v_cvt_i32_f32, v_cvt_f32_i32", 2, d_rec, d_out);
    run<4>("v_cvt_f32_u32 sdwa WORD_0 / WORD_1", 2, d_rec, d_out);
    run<19>("v_cvt_f32_ubyte0 / ubyte2", 2, d_rec, d_out);
    run<20>("a Lanczos4 tap: cvt sdwa, mul, add", 3, d_rec, d_out);
    run<32>("the same without the conversion: mul, add", 2, d_rec, d_out);
    run<24>("mul, max, add, cvt_i32 (2 full-rate + 2 half-rate)", 4, d_rec, d_out);
    run<5>("v_mul_f32 s,v ; v_add_f32 s,v (SGPR operand)", 2, d_rec, d_out);
    run<6>("v_mul_f32, s_add_u32 (both counted)", 2, d_rec, d_out);
    run<26>("v_mul_f32, s_add_u32, s_and_b32 (1 VALU + 2 SALU)", 3, d_rec, d_out);
    run<29>("v_mul_f32, s_and_b64 (both counted)", 2, d_rec, d_out);
    run<16>("v_mul_f32, s_nop 0 (both counted)", 2, d_rec, d_out);
    run<7>("v_cmp vcc + v_cndmask vcc (2 per stage)", 2, d_rec, d_out);
    run<8>("v_cmp_e64 -> sgpr pair + v_cndmask_e64 (2 per stage)", 2, d_rec, d_out);
    run<30>("v_cmp_e64 -> sgpr pair + s_and_b64 (a vote term; 2 per stage)", 2, d_rec, d_out);
    run<9>("v_and_b32, v_add_u32", 2, d_rec, d_out);
    run<10>("v_lshlrev_b32, v_lshrrev_b32", 2, d_rec, d_out);
    run<31>("v_ashrrev_i32, v_bfe_u32", 2, d_rec, d_out);
    run<11>("v_mad_u32_u24, v_mul_u32_u24", 2, d_rec, d_out);
    run<21>("v_alignbit_b32, v_perm_b32", 2, d_rec, d_out);
    run<13>("v_rcp_f32, v_sqrt_f32", 2, d_rec, d_out);
    run<14>("v_pk_mul_f32, v_pk_add_f32 (per packed instruction)", 1, d_rec, d_out);
    run<15>("v_mov_b32, v_mov_b32 dpp quad_perm", 2, d_rec, d_out);
    run<23>("v_readlane_b32, v_add_u32 s,v", 2, d_rec, d_out);
    run<17>("v_mul_f32, ds_read_b32 + v_add_f32 (3 counted)", 3, d_rec, d_out);
    return 0;
}
