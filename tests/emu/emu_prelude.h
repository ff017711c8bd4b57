// emu_prelude.h — TEST INFRASTRUCTURE: what hiprtc's implicit HIP environment gives the fused kernel's source, restated for a HOST build.
//
// tests/emu/ compiles the very text libgfwarp.so embeds for hiprtc (tools/gen_jit_source.py: gfw_frame.hip + its headers, amalgamated) as
// plain C++ for the host cores and interprets the launch lane by lane (emu_driver.cpp), so that the CPU tier of the test suite runs the
// product's kernel SOURCE against the oracle without a GPU.  It is not a fallback: nothing under gyroflow_amd/ knows it exists, the
// library never loads it, and it is as slow as it sounds.  The source text is not edited for this — the only transformation is the
// seven inline-asm statements (AMD mnemonics) turned into calls of the emu_v_* functions below (tests/emu/build_emu.py).
#pragma once
#define GFW_HOST_INTERPRETER 1
#include <math.h>
#include <stddef.h>
#include <stdint.h>
#include <string.h>

#define __device__
#define __host__
#define __global__
#define __forceinline__ inline __attribute__((always_inline))
#define __shared__ static
#define __constant__
#define __launch_bounds__(...)
#define __restrict__

struct float2 { float x, y; };
struct alignas(16) float4 { float x, y, z, w; };
struct uint2 { unsigned x, y; };
struct alignas(16) uint4 { unsigned x, y, z, w; };
struct int2 { int x, y; };
struct dim3 { unsigned x, y, z; dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {} };
// what the launch templates behind the kernels mention (never called here: the interpreter is the launcher)
typedef int hipError_t; typedef void *hipStream_t;
enum { hipSuccess = 0, hipErrorInvalidValue = 1 };
static inline hipError_t hipGetLastError() { return hipSuccess; }
#define hipLaunchKernelGGL(...) ((void)0)
static inline float2 make_float2(float x, float y) { return float2{x, y}; }
static inline float4 make_float4(float x, float y, float z, float w) { return float4{x, y, z, w}; }
static inline uint2 make_uint2(unsigned x, unsigned y) { return uint2{x, y}; }

// ---- the lane being interpreted (emu_driver.cpp switches it) --------------------------------------------------------------
struct EmuLane { dim3 tid, bid, bdim, gdim; };
extern EmuLane *emu_cur;
#define threadIdx (emu_cur->tid)
#define blockIdx (emu_cur->bid)
#define blockDim (emu_cur->bdim)
#define gridDim (emu_cur->gdim)

// ---- synchronisation: the driver runs the 256 lanes of a workgroup as cooperative fibers ----------------------------------
void emu_workgroup_barrier();                 // __syncthreads: every lane of the workgroup
void emu_wave_sync(int site);                 // a wavefront-scope fence: in hardware the wave's lanes are in lockstep around it
#define __syncthreads() emu_workgroup_barrier()
#define __builtin_amdgcn_fence(order, scope) emu_wave_sync(__LINE__)
// wave votes: both outcomes of every vote in this source compute the same bits for a lane (they select a cheaper route when the whole
// wave qualifies), so the lane's own predicate is a valid answer
// (EMU_VOTES = 0, the default).  EMU_VOTES = 1 answers as a wave would in which some OTHER lane fails the test — __all false, __any true — so that
// every lane takes the general route: the two settings together run both sides of each vote over all pixels.
#ifndef EMU_VOTES
#define EMU_VOTES 0
#endif
#if EMU_VOTES == 1
#define __all(p) ((void)(p), 0)
#define __any(p) ((void)(p), 1)
#else
#define __all(p) ((p) ? 1 : 0)
#define __any(p) ((p) ? 1 : 0)
#endif

// A ballot is a real wave operation (the first pass ranks its undecided lanes with it: every lane must see the SAME mask): the wave's live lanes
// rendezvous, the mask is assembled from all of them, they rendezvous again (emu_fibers.inc).  Ballots therefore sit in uniform control flow.
unsigned long long emu_ballot(int pred, int site);
#define __ballot(p) emu_ballot((p) ? 1 : 0, __LINE__)
#define __popcll(x) __builtin_popcountll(x)
static inline unsigned __builtin_amdgcn_mbcnt_lo(unsigned m, unsigned base) { const unsigned l = threadIdx.x; return base + (unsigned)__builtin_popcount(l >= 32 ? m : (m & ((1u << l) - 1u))); }
static inline unsigned __builtin_amdgcn_mbcnt_hi(unsigned m, unsigned base) { const unsigned l = threadIdx.x; return base + (l <= 32 ? 0u : (unsigned)__builtin_popcount(m & ((1u << (l - 32)) - 1u))); }

// A lane reaching an atomic waits until every other live lane of its wave has run as far as it can (to its own atomic, or to a fence): in
// hardware the wave executes the instructions BEFORE the atomic in lockstep, so no lane may see its effect earlier (the first pass reads the
// queue length right after a fence, then lanes push new entries).
void emu_wave_atomic_point();
template <typename T> static inline T atomicAdd(T *p, T v) { emu_wave_atomic_point(); const T old = *p; *p = (T)(old + v); return old; }
// a lane's add into an LDS word of its own (the checksum slots of gfw_frame.hip): nobody else reads it before a fence
#define GFW_LDS_ADD(p, v) ((void)(*(p) += (v)))

// ---- hardware instructions and builtins --------------------------------------------------------------------------------
static inline float emu_sat_i32(float v) { return v; }
static inline int emu_v_cvt_i32_f32(float v) { if (v != v) return 0; if (v >= 2147483648.0f) return INT32_MAX; if (v <= -2147483648.0f) return INT32_MIN; return (int)v; }
static inline uint32_t emu_v_cvt_u32_f32(float v) { if (v != v || v <= 0.0f) return 0u; if (v >= 4294967296.0f) return 0xFFFFFFFFu; return (uint32_t)v; }
static inline float emu_v_min_f32(float a, float b) { return fminf(a, b); }                  // IEEE mode: the non-NaN operand wins
static inline int emu_v_mul_i32_i24(int a, int b) { return (int)((int64_t)((a << 8) >> 8) * (int64_t)((b << 8) >> 8)); }
// v_rcp_f32 / v_sqrt_f32 are 1-ulp approximations; the callers refine them (gfw_fastmath.h).  Here they are the correctly rounded values, moved by
// EMU_HW_ULP units in the last place (0, +1, -1) so that a run can show that the refinement does not depend on WHICH 1-ulp answer the hardware gives.
#ifndef EMU_HW_ULP
#define EMU_HW_ULP 0
#endif
static inline float emu_nudge(float v) {
    if (EMU_HW_ULP == 0 || !(v == v) || v == 0.0f || isinf(v)) return v;
    uint32_t u; memcpy(&u, &v, 4);
    u += (uint32_t)(((u >> 31) ? -1 : 1) * EMU_HW_ULP);          // towards larger magnitude for +n on either sign: a step in the value's own ulp
    memcpy(&v, &u, 4);
    return v;
}
static inline float __builtin_amdgcn_rcpf(float x) { return emu_nudge(1.0f / x); }
static inline float __builtin_amdgcn_sqrtf(float x) { return emu_nudge(sqrtf(x)); }
static inline float __builtin_amdgcn_fractf(float x) { const float f = x - floorf(x); return f < 1.0f ? f : 0x1.fffffep-1f; }
static inline uint32_t __builtin_amdgcn_alignbit(uint32_t hi, uint32_t lo, uint32_t sh) { return (uint32_t)(((((uint64_t)hi) << 32) | lo) >> (sh & 31u)); }
static inline uint32_t __builtin_amdgcn_udot4(uint32_t a, uint32_t b, uint32_t c, bool) {
    uint32_t s = c;
    for (int i = 0; i < 4; ++i) s += ((a >> (8 * i)) & 255u) * ((b >> (8 * i)) & 255u);
    return s;
}
#define __builtin_amdgcn_s_setprio(x) ((void)0)
#define __builtin_amdgcn_s_getreg(x) (0u)
#define __builtin_readcyclecounter() (0ull)

// HIP's integer / float min and max overloads
static inline int min(int a, int b) { return a < b ? a : b; }
static inline int max(int a, int b) { return a > b ? a : b; }
static inline unsigned min(unsigned a, unsigned b) { return a < b ? a : b; }
static inline unsigned max(unsigned a, unsigned b) { return a > b ? a : b; }
static inline long long min(long long a, long long b) { return a < b ? a : b; }
static inline long long max(long long a, long long b) { return a > b ? a : b; }
static inline float min(float a, float b) { return fminf(a, b); }
static inline float max(float a, float b) { return fmaxf(a, b); }
template <typename T> static inline T atomicMax(T *p, T v) { emu_wave_atomic_point(); const T old = *p; if (v > old) *p = v; return old; }
#define amdgpu_waves_per_eu(...)
struct alignas(16) ulonglong2 { unsigned long long x, y; };
// wave shuffles are not interpreted (only the verification checksum kernel uses one): a lane-at-a-time run has no neighbour to read
[[noreturn]] void emu_unsupported(const char *what);
template <typename T> static inline T __shfl_down(T, int, int = 64) { emu_unsupported("__shfl_down"); }
