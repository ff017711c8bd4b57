// membench2.hip — the Lanczos4 tap-row fetch pattern: every lane reads 16 bytes, lanes 4 bytes apart (2 px of u16 each),
// eight consecutive rows, a small L1/L2-resident working set per workgroup.  Which fetch shape does the TA like?
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
typedef unsigned int u4 __attribute__((ext_vector_type(4)));
typedef unsigned int u2 __attribute__((ext_vector_type(2)));

template <int MODE> __global__ __launch_bounds__(256) void k(const unsigned char *src, unsigned *out, int stride, int byte_off, int iters) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    unsigned acc = 0;
    const unsigned char *base = src + (size_t)((blockIdx.x * 4 + wave) % 500) * 16 * stride;     // each wave walks its own band of rows
    for (int it = 0; it < iters; ++it) {
        const unsigned char *p = base + (size_t)(it & 7) * stride + (size_t)((it >> 3) & 15) * 256 + lane * 4 + byte_off;
        #pragma unroll
        for (int r = 0; r < 8; ++r) {
            const unsigned char *q = p + (size_t)r * stride;
            if (MODE == 0) { u4 v; __builtin_memcpy(&v, q, 16); acc += v.x + v.y + v.z + v.w; }                                   // dwordx4 (possibly misaligned)
            if (MODE == 1) { unsigned a, b, c, d; __builtin_memcpy(&a, q, 4); __builtin_memcpy(&b, q + 4, 4); __builtin_memcpy(&c, q + 8, 4); __builtin_memcpy(&d, q + 12, 4);
                             asm volatile("" : "+v"(a), "+v"(b), "+v"(c), "+v"(d)); acc += a + b + c + d; }                        // four dword loads
            if (MODE == 2) { const unsigned *w = (const unsigned *)(q - byte_off); u4 v = *(const u4 *)w; unsigned e = w[4]; const unsigned sh = byte_off * 8;
                             acc += __builtin_amdgcn_alignbit(v.y, v.x, sh) + __builtin_amdgcn_alignbit(v.z, v.y, sh) + __builtin_amdgcn_alignbit(v.w, v.z, sh) + __builtin_amdgcn_alignbit(e, v.w, sh); }
            if (MODE == 3) { u2 v0, v1; __builtin_memcpy(&v0, q, 8); __builtin_memcpy(&v1, q + 8, 8); asm volatile("" : "+v"(v0), "+v"(v1)); acc += v0.x + v0.y + v1.x + v1.y; }   // two dwordx2
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc;
}
template <int MODE> void run(const char *name, const unsigned char *d, unsigned *o, int byte_off) {
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    const int grid = 256 * 6, iters = 256;
    hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(256), 0, 0, d, o, 8192, byte_off, iters);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(256), 0, 0, d, o, 8192, byte_off, iters);
    CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1)); ms /= 5;
    const double rows = (double)grid * 4 * iters * 8;       // wave-level 16-byte-per-lane row fetches
    printf("%-40s offset %d: %7.3f ms  %7.2f G row-fetches/s = one per %5.1f cycles per CU\n", name, byte_off, ms, rows / ms / 1e6, 256.0 * 2.4e9 / (rows / (ms * 1e-3)));
}
int main() {
    unsigned char *d; unsigned *o;
    const size_t bytes = (size_t)(500 * 16 + 64) * 8192;
    CHECK(hipMalloc(&d, bytes)); CHECK(hipMemset(d, 1, bytes)); CHECK(hipMalloc(&o, 256 * 6 * 256 * 4));
    for (int off : {0, 2}) run<0>("one dwordx4", d, o, off);
    for (int off : {0, 2}) run<3>("two dwordx2", d, o, off);
    for (int off : {0, 2}) run<1>("four dword", d, o, off);
    for (int off : {0, 2}) run<2>("aligned dwordx4 + dword, v_alignbit", d, o, off);
    return 0;
}
