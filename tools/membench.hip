// membench.hip — cost of misaligned multi-dword global loads on gfx950 (design data for the Lanczos4 tap fetch).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
typedef unsigned int u4 __attribute__((ext_vector_type(4)));

// each lane reads ROWS x 16 bytes at (row * stride + lane_off + byte_off): the Lanczos4 footprint of a u16 plane
template <int MODE> __global__ __launch_bounds__(256) void k(const unsigned char *src, unsigned *out, int stride, int byte_off, int iters) {
    const int lane = threadIdx.x + blockIdx.x * 256;
    unsigned acc = 0;
    for (int it = 0; it < iters; ++it) {
        const unsigned char *p = src + (size_t)((lane * 37 + it * 8) % 2000) * stride + (size_t)(lane % 448) * 16 + byte_off;
        #pragma unroll
        for (int r = 0; r < 8; ++r) {
            if (MODE == 0) { u4 v; __builtin_memcpy(&v, p + (size_t)r * stride, 16); acc += v.x + v.y + v.z + v.w; }                 // one (mis)aligned dwordx4
            if (MODE == 1) { const unsigned *q = (const unsigned *)(p + (size_t)r * stride - byte_off);                              // aligned dwordx4 + dword, then funnel shift
                             u4 v = *(const u4 *)q; unsigned e = q[4]; const unsigned sh = byte_off * 8;
                             acc += __builtin_amdgcn_alignbit(v.y, v.x, sh) + __builtin_amdgcn_alignbit(v.z, v.y, sh) + __builtin_amdgcn_alignbit(v.w, v.z, sh) + __builtin_amdgcn_alignbit(e, v.w, sh); }
        }
    }
    out[lane] = acc;
}
template <int MODE> void run(const char *name, const unsigned char *d, unsigned *o, int byte_off) {
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    const int grid = 256 * 8, iters = 64;
    hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(256), 0, 0, d, o, 8192, byte_off, iters);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(256), 0, 0, d, o, 8192, byte_off, iters);
    CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1)); ms /= 5;
    const double loads = (double)grid * 256 / 64 * iters * 8;       // wave-level 16-byte-per-lane fetches
    printf("%-44s offset %d: %7.3f ms  %8.2f G wave-fetches/s  (%6.2f TB/s of requested bytes)\n", name, byte_off, ms, loads / ms / 1e6, loads * 1024 / ms / 1e9);
}
int main() {
    unsigned char *d; unsigned *o;
    CHECK(hipMalloc(&d, 2100 * 8192 + 64)); CHECK(hipMemset(d, 1, 2100 * 8192 + 64)); CHECK(hipMalloc(&o, 256 * 8 * 256 * 4));
    for (int off : {0, 2, 4, 6, 8}) run<0>("16 B per lane, one dwordx4 at address+offset", d, o, off);
    for (int off : {0, 2}) run<1>("aligned dwordx4 + dword, v_alignbit", d, o, off);
    return 0;
}
