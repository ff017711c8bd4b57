// lds_taps_probe.hip — would the LUT samplers' taps be cheaper from an LDS tile of FLOATS?  (design data for Lanczos4 / bicubic; not part of the product)
// profiles/r06_tbuffer_probe.txt left the Lanczos4 sampler co-limited by the L1 -> register return path (20 B per lane and tap row at 64 B / clk / CU) and by VALU issue
// (a third of a tap is its integer -> float conversion).  Both fall if a workgroup stages the source window of its tile ONCE: coalesced dword reads, every source
// sample converted once instead of 64 times, floats written to LDS, and the taps read from there (128 B / clk / CU, no alignbit, no conversion).
// This probe walks a 4K plane the way tbuffer_probe does (128 x 16-pixel tiles, a pixel pair per lane, I x I taps per sample, mul / add per tap, no fma) and compares
//   MODE 0  today's fetch (aligned dwords + v_alignbit + v_cvt)                      with
//   MODE 2  bounding box of the tile's windows (wave reductions + LDS), window staged as floats, taps by ds_read
// with the arithmetic and without it.  The sums must be bit-identical.
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off tools/lds_taps_probe.hip -o tools/lds_taps_probe
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
typedef float f4 __attribute__((ext_vector_type(4)));

// the probe's map (tbuffer_probe's): smooth, slightly sheared; SHEAR > 0 tilts it further so that the window grows (rows of the tile move apart)
template <int I>
__device__ __forceinline__ void map_of(int x, int y, int lane, int w, int h, int shift_x, int shear, int &sx, int &sy) {
    sx = x + shift_x + ((y >> 2) & 1) + ((y * shear) >> 4);
    sy = y + ((lane >> 4) & 1) + ((x * shear) >> 6);
    sx = min(max(sx, 0), w - I - 4); sy = min(max(sy, 0), h - I);
}

template <typename T, int I, int ARITH>
__global__ __launch_bounds__(256) void k_taps(const uint8_t *src, int stride, int w, int h, float *out, const float *lut, int shift_x, int shear) {
    __shared__ float s_lut[448];
    for (int i = threadIdx.x + threadIdx.y * 64; i < 448; i += 256) s_lut[i] = lut[i];
    __syncthreads();
    const int tiles_x = w / 128;
    const int tx = blockIdx.x % tiles_x, ty = blockIdx.x / tiles_x;
    const int lane = threadIdx.x, wave = threadIdx.y;
    float acc = 0.0f; uint32_t xacc = 0;
    #pragma unroll 1
    for (int r = 0; r < 4; ++r) {
        const int y = ty * 16 + wave * 4 + r;
        #pragma unroll 1
        for (int px = 0; px < 2; ++px) {
            const int x = tx * 128 + lane * 2 + px;
            int sx, sy; map_of<I>(x, y, lane, w, h, shift_x, shear, sx, sy);
            const float *cx = s_lut + 192 + ((x * 5) & 31) * 8, *cy = s_lut + 192 + ((y * 3) & 31) * 8;
            const uint32_t off0 = (uint32_t)sy * (uint32_t)stride + (uint32_t)sx * (uint32_t)sizeof(T);
            float s1 = 0.0f;
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
            acc += s1;
        }
    }
    out[(size_t)blockIdx.x * 256 + wave * 64 + lane] = acc + (ARITH ? 0.0f : (float)(xacc & 0xffffu));
}

// ---- the staged form -------------------------------------------------------------------------------------------------------------------------------------
// PITCH floats per staged row, ROWS rows: the tile's budget (a tile whose window does not fit would take today's path in a product kernel; here it is counted).
// RD: 0 = let the compiler choose the LDS reads (4-byte aligned float loads), 1 = ds_read_b128 at 4-byte alignment (inline asm; needs the unaligned access mode ROCm sets)
template <int I, int PITCH, int RD>
__device__ __forceinline__ void lds_window(const float *p, float (*t)[8]) {
    if (RD == 1 && I >= 4) {
        const uint32_t a = (uint32_t)(uintptr_t)p;                          // LDS byte address (the low 32 bits of a __shared__ pointer)
        // the compiler's wait-count pass does not see LDS reads inside inline asm: the block waits for its own reads
        if (I == 8) {
            f4 q[16];
            asm volatile(
                "ds_read_b128 %0, %16 offset:%c17\n\tds_read_b128 %1, %16 offset:%c18\n\tds_read_b128 %2, %16 offset:%c19\n\tds_read_b128 %3, %16 offset:%c20\n\t"
                "ds_read_b128 %4, %16 offset:%c21\n\tds_read_b128 %5, %16 offset:%c22\n\tds_read_b128 %6, %16 offset:%c23\n\tds_read_b128 %7, %16 offset:%c24\n\t"
                "ds_read_b128 %8, %16 offset:%c25\n\tds_read_b128 %9, %16 offset:%c26\n\tds_read_b128 %10, %16 offset:%c27\n\tds_read_b128 %11, %16 offset:%c28\n\t"
                "ds_read_b128 %12, %16 offset:%c29\n\tds_read_b128 %13, %16 offset:%c30\n\tds_read_b128 %14, %16 offset:%c31\n\tds_read_b128 %15, %16 offset:%c32\n\t"
                "s_waitcnt lgkmcnt(0)"
                : "=&v"(q[0]), "=&v"(q[1]), "=&v"(q[2]), "=&v"(q[3]), "=&v"(q[4]), "=&v"(q[5]), "=&v"(q[6]), "=&v"(q[7]),
                  "=&v"(q[8]), "=&v"(q[9]), "=&v"(q[10]), "=&v"(q[11]), "=&v"(q[12]), "=&v"(q[13]), "=&v"(q[14]), "=&v"(q[15])
                : "v"(a), "n"(0 * PITCH * 4), "n"(0 * PITCH * 4 + 16), "n"(1 * PITCH * 4), "n"(1 * PITCH * 4 + 16), "n"(2 * PITCH * 4), "n"(2 * PITCH * 4 + 16),
                  "n"(3 * PITCH * 4), "n"(3 * PITCH * 4 + 16), "n"(4 * PITCH * 4), "n"(4 * PITCH * 4 + 16), "n"(5 * PITCH * 4), "n"(5 * PITCH * 4 + 16),
                  "n"(6 * PITCH * 4), "n"(6 * PITCH * 4 + 16), "n"(7 * PITCH * 4), "n"(7 * PITCH * 4 + 16)
                : "memory");
            #pragma unroll
            for (int yp = 0; yp < 8; ++yp) { t[yp][0] = q[2 * yp].x; t[yp][1] = q[2 * yp].y; t[yp][2] = q[2 * yp].z; t[yp][3] = q[2 * yp].w;
                                             t[yp][4] = q[2 * yp + 1].x; t[yp][5] = q[2 * yp + 1].y; t[yp][6] = q[2 * yp + 1].z; t[yp][7] = q[2 * yp + 1].w; }
        } else {
            f4 q[4];
            asm volatile(
                "ds_read_b128 %0, %4 offset:%c5\n\tds_read_b128 %1, %4 offset:%c6\n\tds_read_b128 %2, %4 offset:%c7\n\tds_read_b128 %3, %4 offset:%c8\n\ts_waitcnt lgkmcnt(0)"
                : "=&v"(q[0]), "=&v"(q[1]), "=&v"(q[2]), "=&v"(q[3])
                : "v"(a), "n"(0 * PITCH * 4), "n"(1 * PITCH * 4), "n"(2 * PITCH * 4), "n"(3 * PITCH * 4)
                : "memory");
            #pragma unroll
            for (int yp = 0; yp < 4; ++yp) { t[yp][0] = q[yp].x; t[yp][1] = q[yp].y; t[yp][2] = q[yp].z; t[yp][3] = q[yp].w; }
        }
    } else {
        #pragma unroll
        for (int yp = 0; yp < I; ++yp) {
            #pragma unroll
            for (int j = 0; j < I; ++j) t[yp][j] = p[yp * PITCH + j];
        }
    }
}

template <typename T, int I, int ARITH, int PITCH, int ROWS, int RD>
__global__ __launch_bounds__(256) void k_taps_lds(const uint8_t *src, int stride, int w, int h, float *out, const float *lut, int shift_x, int shear, unsigned *misfit) {
    __shared__ float s_lut[448];
    __shared__ int s_box[4][4];
    __shared__ __attribute__((aligned(16))) float s_win[ROWS * PITCH];
    for (int i = threadIdx.x + threadIdx.y * 64; i < 448; i += 256) s_lut[i] = lut[i];
    const int tiles_x = w / 128;
    const int tx = blockIdx.x % tiles_x, ty = blockIdx.x / tiles_x;
    const int lane = threadIdx.x, wave = threadIdx.y;
    // (1) the bounding box of the tile's windows: per lane over its 8 samples, per wave by butterfly, per workgroup through LDS
    int x0 = 0x7fffffff, x1 = -1, y0 = 0x7fffffff, y1 = -1;
    #pragma unroll
    for (int r = 0; r < 4; ++r) {
        #pragma unroll
        for (int px = 0; px < 2; ++px) {
            int sx, sy; map_of<I>(tx * 128 + lane * 2 + px, ty * 16 + wave * 4 + r, lane, w, h, shift_x, shear, sx, sy);
            x0 = min(x0, sx); x1 = max(x1, sx); y0 = min(y0, sy); y1 = max(y1, sy);
        }
    }
    #pragma unroll
    for (int m = 1; m < 64; m <<= 1) {
        x0 = min(x0, __shfl_xor(x0, m)); x1 = max(x1, __shfl_xor(x1, m)); y0 = min(y0, __shfl_xor(y0, m)); y1 = max(y1, __shfl_xor(y1, m));
    }
    if (lane == 0) { s_box[wave][0] = x0; s_box[wave][1] = x1; s_box[wave][2] = y0; s_box[wave][3] = y1; }
    __syncthreads();
    x0 = min(min(s_box[0][0], s_box[1][0]), min(s_box[2][0], s_box[3][0])); x1 = max(max(s_box[0][1], s_box[1][1]), max(s_box[2][1], s_box[3][1]));
    y0 = min(min(s_box[0][2], s_box[1][2]), min(s_box[2][2], s_box[3][2])); y1 = max(max(s_box[0][3], s_box[1][3]), max(s_box[2][3], s_box[3][3]));
    constexpr int PER = 8 / (int)sizeof(T);                  // samples per 8-byte fetch
    x0 &= ~(PER - 1);                                        // the window starts on an 8-byte boundary of the source row (planes and strides are 8-byte multiples here)
    const int nx = x1 + I - x0, ny = y1 + I - y0;            // samples per row, rows
    const int nq = (nx + PER - 1) / PER;                     // 8-byte fetches per row
    const bool fits = nq * PER <= PITCH && ny <= ROWS;
    if (!fits) { if (lane == 0 && wave == 0) atomicAdd(misfit, 1u); out[(size_t)blockIdx.x * 256 + wave * 64 + lane] = 0.0f; return; }
    // (2) stage: wave v takes rows v, v + 4, ...; lane l the l-th 8-byte group of the row (nq <= 64).  All fetches of a wave leave before the first conversion.
    {
        constexpr int RPW = (ROWS + 3) / 4;
        uint2 raw[RPW];
        #pragma unroll
        for (int k = 0; k < RPW; ++k) {
            const int row = wave + 4 * k;
            raw[k] = make_uint2(0u, 0u);
            if (row < ny && lane < nq) raw[k] = *reinterpret_cast<const uint2 *>(src + (size_t)(y0 + row) * (size_t)stride + (size_t)x0 * sizeof(T) + (size_t)lane * 8);
        }
        #pragma unroll
        for (int k = 0; k < RPW; ++k) {
            const int row = wave + 4 * k;
            if (row < ny && lane < nq) {
                float *d = s_win + row * PITCH + lane * PER;
                if (sizeof(T) == 2) {
                    f4 v; v.x = (float)(raw[k].x & 0xffffu); v.y = (float)(raw[k].x >> 16); v.z = (float)(raw[k].y & 0xffffu); v.w = (float)(raw[k].y >> 16);
                    *reinterpret_cast<f4 *>(d) = v;
                } else {
                    f4 v0, v1;
                    v0.x = (float)(raw[k].x & 0xffu); v0.y = (float)((raw[k].x >> 8) & 0xffu); v0.z = (float)((raw[k].x >> 16) & 0xffu); v0.w = (float)(raw[k].x >> 24);
                    v1.x = (float)(raw[k].y & 0xffu); v1.y = (float)((raw[k].y >> 8) & 0xffu); v1.z = (float)((raw[k].y >> 16) & 0xffu); v1.w = (float)(raw[k].y >> 24);
                    *reinterpret_cast<f4 *>(d) = v0; *reinterpret_cast<f4 *>(d + 4) = v1;
                }
            }
        }
    }
    __syncthreads();
    // (3) the taps, from the window
    float acc = 0.0f; uint32_t xacc = 0;
    #pragma unroll 1
    for (int r = 0; r < 4; ++r) {
        const int y = ty * 16 + wave * 4 + r;
        #pragma unroll 1
        for (int px = 0; px < 2; ++px) {
            const int x = tx * 128 + lane * 2 + px;
            int sx, sy; map_of<I>(x, y, lane, w, h, shift_x, shear, sx, sy);
            const float *cx = s_lut + 192 + ((x * 5) & 31) * 8, *cy = s_lut + 192 + ((y * 3) & 31) * 8;
            const float *p = s_win + (sy - y0) * PITCH + (sx - x0);
            float s1 = 0.0f;
            float t[I][8];
            lds_window<I, PITCH, RD>(p, t);
            #pragma unroll
            for (int yp = 0; yp < I; ++yp) {
                if (ARITH) {
                    float xs = t[yp][0] * cx[0];
                    #pragma unroll
                    for (int j = 1; j < I; ++j) xs = xs + t[yp][j] * cx[j];
                    s1 = s1 + xs * cy[yp];
                } else {
                    #pragma unroll
                    for (int j = 0; j < I; ++j) xacc ^= __builtin_bit_cast(uint32_t, t[yp][j]);
                }
            }
            acc += s1;
        }
    }
    out[(size_t)blockIdx.x * 256 + wave * 64 + lane] = acc + (ARITH ? 0.0f : (float)(xacc & 0xffffu));
}

static double time_it(const char *name, void (*launch)(void *), void *ctx, float *d_out, size_t n_out, std::vector<float> *res) {
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    for (int i = 0; i < 3; ++i) launch(ctx);
    CHECK(hipDeviceSynchronize());
    const int reps = 20;
    CHECK(hipEventRecord(e0));
    for (int i = 0; i < reps; ++i) launch(ctx);
    CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    if (res) { res->resize(n_out); CHECK(hipMemcpy(res->data(), d_out, n_out * 4, hipMemcpyDeviceToHost)); }
    printf("  %-78s %8.2f us per plane pass\n", name, ms * 1e3 / reps);
    CHECK(hipEventDestroy(e0)); CHECK(hipEventDestroy(e1));
    return ms * 1e3 / reps;
}

struct Ctx { const uint8_t *src; int stride, w, h; float *out; const float *lut; int shear; unsigned *misfit; };
template <typename T, int I, int ARITH> static void l_direct(void *c_) { Ctx *c = (Ctx *)c_;
    hipLaunchKernelGGL((k_taps<T, I, ARITH>), dim3((c->w / 128) * (c->h / 16)), dim3(64, 4), 0, 0, c->src, c->stride, c->w, c->h, c->out, c->lut, 3, c->shear); }
template <typename T, int I, int ARITH, int PITCH, int ROWS, int RD> static void l_lds(void *c_) { Ctx *c = (Ctx *)c_;
    hipLaunchKernelGGL((k_taps_lds<T, I, ARITH, PITCH, ROWS, RD>), dim3((c->w / 128) * (c->h / 16)), dim3(64, 4), 0, 0, c->src, c->stride, c->w, c->h, c->out, c->lut, 3, c->shear, c->misfit); }

static bool same(const std::vector<float> &a, const std::vector<float> &b) { return a.size() == b.size() && memcmp(a.data(), b.data(), a.size() * 4) == 0; }

template <typename T, int I, int PITCH, int ROWS>
static void series(const char *what, Ctx &c) {
    std::vector<float> a, b, d;
    const size_t n_out = (size_t)(c.w / 128) * (c.h / 16) * 256;
    char name[160];
    unsigned mf;
    snprintf(name, sizeof name, "%s, today's fetch (dwords + alignbit + cvt), mul/add", what);
    const double t0 = time_it(name, l_direct<T, I, 1>, &c, c.out, n_out, &a);
    CHECK(hipMemset(c.misfit, 0, 4));
    snprintf(name, sizeof name, "%s, window staged as floats (%d x %d), compiler's LDS reads, mul/add", what, PITCH, ROWS);
    const double t1 = time_it(name, l_lds<T, I, 1, PITCH, ROWS, 0>, &c, c.out, n_out, &b);
    CHECK(hipMemcpy(&mf, c.misfit, 4, hipMemcpyDeviceToHost));
    printf("    sums bit-identical: %s   tiles whose window did not fit: %u of %zu launches x tiles   ratio %.2f\n", same(a, b) ? "yes" : "NO", mf, (size_t)23 * (n_out / 256), t1 / t0);
    snprintf(name, sizeof name, "%s, window staged as floats, ds_read_b128 at 4-byte alignment, mul/add", what);
    const double t2 = time_it(name, l_lds<T, I, 1, PITCH, ROWS, 1>, &c, c.out, n_out, &d);
    printf("    sums bit-identical: %s   ratio %.2f\n", same(a, d) ? "yes" : "NO", t2 / t0);
    snprintf(name, sizeof name, "%s, today's fetch, FETCH ONLY", what);
    time_it(name, l_direct<T, I, 0>, &c, c.out, n_out, nullptr);
    snprintf(name, sizeof name, "%s, staged, compiler's LDS reads, FETCH ONLY", what);
    time_it(name, l_lds<T, I, 0, PITCH, ROWS, 0>, &c, c.out, n_out, nullptr);
    snprintf(name, sizeof name, "%s, staged, ds_read_b128, FETCH ONLY", what);
    time_it(name, l_lds<T, I, 0, PITCH, ROWS, 1>, &c, c.out, n_out, nullptr);
}

int main() {
    const int w = 3840, h = 2160;
    std::vector<uint16_t> h16((size_t)w * h); std::vector<uint8_t> h8((size_t)w * h);
    uint32_t s = 0x9F10u;
    for (size_t i = 0; i < h16.size(); ++i) { s = s * 1664525u + 1013904223u; h16[i] = (uint16_t)(s >> 16); h8[i] = (uint8_t)(s >> 8); }
    h16[0] = 65535; h16[1] = 0; h16[2] = 32768; h8[0] = 255; h8[1] = 0;
    uint16_t *d16; uint8_t *d8; unsigned *d_mis; float *d_out, *d_lut;
    CHECK(hipMalloc(&d16, h16.size() * 2 + 4096)); CHECK(hipMalloc(&d8, h8.size() + 4096)); CHECK(hipMalloc(&d_mis, 64)); CHECK(hipMalloc(&d_out, (size_t)(w / 128) * (h / 16) * 256 * 4)); CHECK(hipMalloc(&d_lut, 448 * 4));
    CHECK(hipMemcpy(d16, h16.data(), h16.size() * 2, hipMemcpyHostToDevice)); CHECK(hipMemcpy(d8, h8.data(), h8.size(), hipMemcpyHostToDevice));
    std::vector<float> lut(448); for (int i = 0; i < 448; ++i) lut[i] = 0.001f * (float)((i * 37) % 211) - 0.05f;
    CHECK(hipMemcpy(d_lut, lut.data(), 448 * 4, hipMemcpyHostToDevice));
    Ctx c16{(const uint8_t *)d16, w * 2, w, h, d_out, d_lut, 0, d_mis}, c8{d8, w, w, h, d_out, d_lut, 0, d_mis};
    for (int shear : {0, 1}) {
        c16.shear = c8.shear = shear;
        printf("u16 plane 3840 x 2160, one sample per pixel, shear %d:\n", shear);
        series<uint16_t, 8, 152, 28>("Lanczos4 8x8", c16);
        series<uint16_t, 4, 144, 24>("bicubic 4x4", c16);
        series<uint16_t, 2, 144, 22>("bilinear 2x2", c16);
        printf("u8 plane 3840 x 2160, shear %d:\n", shear);
        series<uint8_t, 8, 152, 28>("Lanczos4 8x8", c8);
        series<uint8_t, 4, 144, 24>("bicubic 4x4", c8);
    }
    return 0;
}
