// fetch_calib.hip — what do FETCH_SIZE / WRITE_SIZE report for THIS library's access shapes?  (round-5 verdict, weak #9)
// MI355X_MICROARCH.md (HBM): on gfx950 FETCH_SIZE reads exactly half the bytes of a WIDE coalesced streaming read (16 B per lane); "other access widths and WRITE_SIZE
// are uncalibrated: calibrate on a known byte count in your own access pattern".  The warp kernels read dwords (a pixel pair of u16, lanes 4 bytes apart) and store
// dwords (luma pair) and shorts (a chroma sample per lane).  Each kernel below moves a KNOWN number of bytes exactly once over a buffer far larger than the
// Infinity Cache (1 GiB against 256 MiB), so bytes / counter is the factor for that shape.  Run under rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE (separate passes).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
typedef unsigned int u4 __attribute__((ext_vector_type(4)));

// reads: every byte of the buffer once
__global__ __launch_bounds__(256) void calib_read_dwordx4(const u4 *src, unsigned *out, size_t n16) {
    unsigned acc = 0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256) { const u4 v = src[i]; acc += v.x ^ v.y ^ v.z ^ v.w; }
    if (acc == 0x12345u) out[0] = acc;
}
__global__ __launch_bounds__(256) void calib_read_dword(const unsigned *src, unsigned *out, size_t n4) {
    unsigned acc = 0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) acc += src[i];
    if (acc == 0x12345u) out[0] = acc;
}
// the bilinear tap shape: two rows, a dword per lane at a 2-byte-aligned (every other lane: misaligned) address, lanes 4 bytes apart — each wave instruction covers
// 256 + 2 bytes; the rows of one "plane" are read top to bottom once (row r by iteration r), so every line is fetched from memory once
__global__ __launch_bounds__(256) void calib_read_taps(const uint8_t *src, unsigned *out, int stride, int rows) {
    unsigned acc = 0;
    const int col = (blockIdx.x * 256 + threadIdx.x) * 4 + ((threadIdx.x & 1) ? 2 : 0);
    if (col + 4 > stride) return;
    for (int r = 0; r + 1 < rows; ++r) {
        unsigned a, b;
        __builtin_memcpy(&a, src + (size_t)r * stride + col, 4); __builtin_memcpy(&b, src + (size_t)(r + 1) * stride + col, 4);
        acc += a ^ b;
    }
    if (acc == 0x12345u) out[0] = acc;
}
// writes: every byte of the buffer once
__global__ __launch_bounds__(256) void calib_write_dwordx4(u4 *dst, size_t n16) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256) dst[i] = u4{(unsigned)i, 1u, 2u, 3u};
}
__global__ __launch_bounds__(256) void calib_write_dword(unsigned *dst, size_t n4) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) dst[i] = (unsigned)i;
}
__global__ __launch_bounds__(256) void calib_write_short(uint16_t *dst, size_t n2) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n2; i += (size_t)gridDim.x * 256) dst[i] = (uint16_t)i;
}

int main() {
    const size_t bytes = 1ull << 30;
    uint8_t *buf; unsigned *out;
    CHECK(hipMalloc(&buf, bytes)); CHECK(hipMalloc(&out, 64)); CHECK(hipMemset(buf, 1, bytes)); CHECK(hipDeviceSynchronize());
    const int grid = 256 * 8;
    hipLaunchKernelGGL(calib_read_dwordx4, dim3(grid), dim3(256), 0, 0, (const u4 *)buf, out, bytes / 16); CHECK(hipDeviceSynchronize());
    hipLaunchKernelGGL(calib_read_dword, dim3(grid), dim3(256), 0, 0, (const unsigned *)buf, out, bytes / 4); CHECK(hipDeviceSynchronize());
    {   // 1 GiB as a plane of 128 KiB rows x 8192 rows; one lane per 4 bytes of a row
        const int stride = 128 * 1024, rows = (int)(bytes / stride);
        hipLaunchKernelGGL(calib_read_taps, dim3(stride / 4 / 256), dim3(256), 0, 0, buf, out, stride, rows); CHECK(hipDeviceSynchronize());
    }
    hipLaunchKernelGGL(calib_write_dwordx4, dim3(grid), dim3(256), 0, 0, (u4 *)buf, bytes / 16); CHECK(hipDeviceSynchronize());
    hipLaunchKernelGGL(calib_write_dword, dim3(grid), dim3(256), 0, 0, (unsigned *)buf, bytes / 4); CHECK(hipDeviceSynchronize());
    hipLaunchKernelGGL(calib_write_short, dim3(grid), dim3(256), 0, 0, (uint16_t *)buf, bytes / 2); CHECK(hipDeviceSynchronize());
    printf("fetch_calib: each kernel moved %zu bytes once\n", bytes);
    return 0;
}
