// gfw_frame.hip — fused YUV frame kernel: all planes of one frame in ONE launch.
//
// The reference warps a frame plane by plane (src/rendering/mod.rs:655-658), re-deriving the source
// coordinate of every chroma site from scratch although — with its own arithmetic — the chroma site (xc, yc)
// of a plane subsampled by (DW, DH) evaluates `undistort_coord` at exactly the luma position (DW*xc, DH*yc):
//   map_coord(xc, 0, ow/DW, 0, ow) = xc*ow/(ow/DW) = DW*xc       (exact in f32 when xc*ow is exact; checked on host)
// So one lane owns DW x DH luma pixels plus the chroma site that shares the block's top-left coordinate:
// undistort_coord runs once per luma pixel and never for chroma (2x fewer evaluations for 4:2:2, 1.5x for
// 4:2:0, 3x for 4:4:4), U and V share bins, weights and offsets, and the per-plane source_rect map
// (cpu_undistort.rs:511-514) is a validated divide-by-constant.
//
// Arithmetic is the reference's operation sequence (cpu_undistort.rs:133-228, :421-517, :371-418,
// opencv_fisheye.rs:72-95) with the range scaffolding of divide/sqrt removed (gfw_fastmath.h); operands outside
// the proven range take the generic IEEE path in a (practically never taken) side branch.  Output is
// bit-identical to running gfw_plane_kernel once per plane; tests/test_gpu_parity.py checks both against the
// oracle.
//
// Shape of the kernel (all of it driven by measurements in profiles/):
//   * VALU-issue bound (86 % VALU-active in the shipped configuration), so instruction count and code size are what matter: every code
//     path exists once (row loop not unrolled) and slow paths are side branches — a fully unrolled 26 K-instruction body measured slower
//     (instruction cache); uniforms pinned in VGPRs, LDS copies of the matrix rows and of the source window, a certified second pass,
//     dynamic work distribution: each built, validated bit-exact, measured slower and removed (profiles/r02_*, r03_ab_*.txt);
//   * persistent workgroups: the grid is sized to the machine and each workgroup walks a band of tiles, so the
//     ~5 us start-up of a wave (kernel-argument loads) is paid once, not once per tile; the issue priority of a wave follows the
//     work it has left (GFW_PRIO_MODE), which keeps the waves of a SIMD level;
//   * XCD-banded tile order (workgroups b, b + 8, ... share an XCD): an XCD's L2 sees contiguous runs of source lines — four sub-bands of a
//     frame per XCD, rotated from frame to frame of a clip launch, so that no XCD always gets the cheap (top, bottom) or the dear regions;
//   * two builds of one source: ahead of time, every instantiation the dispatcher can reach (arguments in the kernel-argument
//     segment); and at run time, per clip, ONE instantiation with the clip's constants as literals and up to 16 frames per launch
//     (GFW_JIT / GFW_BAKE, gfw_jit.hip) — 74 against 51 us per 4K frame.
//
// Eligibility (decided on the host, gfw_api.hip build_yuv_args): bilinear / bicubic / Lanczos4 taps (this file is
// compiled once per tap count and sample type), background_mode 0-3, input rotation and the fill flag (round 4; not the colour-range fix), translation3d == 0 (other lens models, refraction, digital lens, IBIS/OIS terms, the lens-correction
// blend, background mode 3 and the Sony mesh are served by the generic-model instantiations with the exact first pass), any stretch (round 4),
// full-plane rects; Luma8/Luma16 (+UV8/UV16) planes with chroma planes of identical geometry, one
// packed RGB(A)8/16 / BGRA8 / AYUV16 / RGBAf plane, or planar R32f planes.
#ifndef GFW_JIT
#define GFW_JIT 0                // 1: this file is being compiled at run time by hiprtc (gfw_jit.hip) into ONE baked instantiation: device code only
#endif
#if !GFW_JIT
#include <hip/hip_runtime.h>
#endif
#include "gfw_warp.h"
#include "gfw_fastmath.h"
#include "gfw_frame.h"
#if !GFW_JIT
#include <cstdio>
#include <cstdlib>
#endif

#ifndef GFW_WAVES_PER_EU
#define GFW_WAVES_PER_EU 6       // register budget of the AHEAD-OF-TIME frame kernels, in waves per SIMD (512 / N VGPRs).  6: 64-71 VGPRs, 106 SGPRs = six
                                 // workgroups per CU.  With the arguments in SGPRs 7 (94 SGPRs, more scalar reloads) measured 2-4 % slower and 8 (78 SGPRs:
                                 // 700 v_readlane) 6 % slower: profiles/r02_scheduling_experiments.md.  The baked builds need far fewer SGPRs and run at
                                 // 8 or 7 (GFW_JIT_WAVES, gfw_api.hip jit_waves)
#endif
#ifndef GFW_HOT_ONLY
#define GFW_HOT_ONLY 0           // A/B builds (tools/build_variants.sh): only the C2 instantiation (u16, 4:2:2 planar, bilinear), seconds to compile
#endif
#ifndef GFW_GENERIC_WAVES_PER_EU
// register budget of the generic-model instantiations (every other lens model, digital lenses, refraction, IBIS/OIS, lens-correction
// blend, background mode 3, Sony mesh), in waves per SIMD.  Bilinear: 6 (80 VGPRs, 0.2-1.4 KB of scratch per lane) — a SuperView
// clip runs 270.3 us at 3, 266.0 at 4, 225.6 at 6, 352.8 at 2 (profiles/r03_ab_northstar.txt).  Bicubic / Lanczos4: 3, not re-measured.
#define GFW_GENERIC_WAVES_PER_EU (GFW_FRAME_TAPS == 2 ? 6 : 3)
#endif
#ifndef GFW_PRIO_MODE
#define GFW_PRIO_MODE 1          // wave issue priority by remaining work (s_setprio).  The SIMD arbiter serves the oldest wave first, so the
                                 // waves of a SIMD progress at 0.115 ... 0.196 lane-rows/us and finish up to 17 us apart
                                 // (profiles/r02_wave_timeline.txt).  1: the priority steps 3 -> 0 as the wave's remaining lane-rows fall below
                                 // 3, 2 and 1 x (its total / GFW_PRIO_SPAN), re-evaluated every tile (every row until round 4: ~20 scalar instructions per row, 46.4 -> 46.15 us): waves with more work left are served first
                                 // and progress stays level.  0: off.  What it is worth depends on how long a wave lives: nothing on a lone 4K
                                 // frame at 6 waves (79.3 = 79.3 us), 4 us of 59 once a launch carries 8 frames at 8 waves per SIMD.
#endif
#ifndef GFW_PRIO_SPAN
#define GFW_PRIO_SPAN 6          // round 3, C2, 8 waves, 63 lane-rows per wave: fixed divisors 3 / 4 / 5 / 6 / 8 / 12 / 16 / 24 / 32 / 64 gave
                                 // 57.3 / 56.4 / 56.0 / 55.5 / 55.0 / 55.1 / 55.6 / 56.4 / 56.9 / 58.4 us; 1080p (18 rows per wave) wants 3, 8K
                                 // (253 rows) 16 or more: the step is a sixth of the wave's work (profiles/r03_ab_waves_priority.txt)
#endif
// Tap rows of a bicubic / Lanczos4 sample in flight (single-channel integer planes, taps_inside): more rows = more fetches outstanding, and more registers.  Round 4's last
// scans (profiles/r04_ab_lut_rows.txt): at 8 waves per SIMD (64 VGPRs) two rows fit and four spill; at 6 waves four fit, at 5 all eight of an 8-bit Lanczos4 sample —
// bicubic 65.5 -> 50.4 us (NV12), 68.5 -> 52.7 (P010); Lanczos4 107.1 -> 91.9 (NV12), 122.4 -> 112.5 (P010), C2 unchanged (142).  The host picks the waves
// (gfw_api.hip jit_waves), the rows follow from them; ahead of time (6 waves, more live scalars) four rows, two for 16-bit Lanczos4 (four spill there).
#ifdef GFW_TAP_ROWS_FORCE
#define GFW_TAP_ROW_UNROLL(I, T) (GFW_TAP_ROWS_FORCE)
#elif GFW_BAKE
#define GFW_TAP_ROW_UNROLL(I, T) (GFW_JIT_WAVES >= 7 ? 2 : ((I) == 4 ? 4 : (GFW_JIT_WAVES <= 5 ? 8 : 4)))
#else
#define GFW_TAP_ROW_UNROLL(I, T) ((I) == 4 ? 4 : (sizeof(T) == 1 ? 4 : 2))
#endif
#ifndef GFW_ROW_CLUSTER
#define GFW_ROW_CLUSTER 0       // measured in round 5 (46.4 = 46.4 us; with the row's stores held back behind the NEXT row's matrix fetches 44.3 against 42.3: profiles/r05_ab_ablations.txt): the luma pair's and the chroma site's taps of a
                                 // 4:2:2 / 4:4:4 planar lane-row fetched in ONE cluster, stores last (the luma store between them is an aliasing barrier: the row waits twice)
#endif
#ifndef GFW_P1_LATTICE
#define GFW_P1_LATTICE 1         // round 5: the certified first pass evaluated at the nodes of a lattice (every 8th luma column of the wave's first and last row: one node per
                                 // lane, once per tile) and interpolated bilinearly for the pixels, its curvature added to the certificate's half-width (DESIGN.md
                                 // section 2c); 0: evaluated per pixel (round 2-4; still the path of clips with an r-limit, whose test is per pixel)
#endif
#ifndef GFW_P1_LATTICE_MAX_E
#define GFW_P1_LATTICE_MAX_E 0.04f
#endif
#ifndef GFW_FASTROW
#define GFW_FASTROW 1            // the branch-free lane-row of phase 3 (rd_lean_nobranch + one `__any` / `__all` per stage); 0: the per-pixel divergent code only (A/B)
#endif
#ifndef GFW_BAKE
#define GFW_BAKE 0               // 1: the clip-invariant arguments are the literals GFW_BK_<field> of the bake header in front of this file (read through AF())
#endif
#ifndef GFW_TLB
#define GFW_TLB(k) do {} while (0)
#define GFW_TLB_START() do {} while (0)
#endif
#ifndef GFW_TIMELINE
#define GFW_TIMELINE 0           // diagnosis builds only: per-wave start / end / phase clocks and HW_ID into a device array that the 60th launch
                                 // dumps to $GFW_TIMELINE_FILE (tools/analyze_timeline.py)
#endif

// Template value of MODEL for the generic-model instantiation that also carries background mode 3 (margin with feather) and the
// Sony mesh / focal-plane-distortion terms: kept apart so that the plain generic instantiation (-1) keeps its register budget.
#define GFW_MODEL_GENERIC_EXTRA (-2)
// A clip-invariant field of the argument block: the argument itself, or — in a baked build — its literal from the bake header.
#if GFW_BAKE
#define AF(x) (GFW_BK_##x)
#define AFA(x, i) (GFW_BK_##x##_##i)              // element i of an array field
#define AFM(m, f) (GFW_BK_##m##_##f)              // member f of a map-constant field
#define GFW_BAKED_DIGITAL ((GFW_BK_extras & 2) != 0)   // a digital lens rides on the specialised fisheye projection (baked builds only)
#define GFW_BAKED_DIGITAL_MODEL (((GFW_BK_extras & 2) != 0) ? GFW_BK_digital : -1)     // the generic branch's digital lens as a literal
#define GFW_CLIP_DIGITAL (GFW_BK_digital)
#else
#define AF(x) (A.x)
#define AFA(x, i) (A.x[i])
#define AFM(m, f) (A.m.f)
#define GFW_BAKED_DIGITAL false
#define GFW_BAKED_DIGITAL_MODEL (-1)
#define GFW_CLIP_DIGITAL (A.common.digital)
#endif

// Timing ablations (wrong output by design) exist only in builds that say GFW_TESTING=1 — the A/B builds of tools/ through GFW_JIT_DEFS, from a library whose embedded
// source was generated with GFW_TESTING_SOURCE=1 (tools/gen_jit_source.py).  The shipped library and every kernel it compiles at run time carry none: the generator replaces
// what lies between the GFW-TESTING markers by the two folded definitions, so GFW_ABL is 0 whatever GFW_JIT_DEFS says — a drop-in library is not one integer or one
// environment variable away from wrong frames (round-5 verdict, weak #10).
// [GFW-TESTING-BEGIN]
#ifndef GFW_TESTING
#define GFW_TESTING 0
#endif
#if GFW_TESTING
#define GFW_ABL(bits) (AF(ablate) & (bits))
#else
#define GFW_ABL(bits) (0)
#endif
#if GFW_BAKE && GFW_TESTING && defined(GFW_ABLATE_FORCE)
#undef GFW_BK_ablate                 // timing ablations of a SPECIALISED kernel (GFW_JIT_DEFS="GFW_TESTING=1;GFW_ABLATE_FORCE=<bits>"): wrong output by design
#define GFW_BK_ablate (GFW_ABLATE_FORCE)
#endif
// [GFW-TESTING-END]

// ---- the frame's checksum, taken where the pixels leave (specialised builds with GFW_BK_checksum: gfw_set_frame_checksums) -----------------------------
// gfw_checksum64 of a destination buffer is the sum of its little-endian 64-bit words modulo 2^64, i.e. every byte times 256^(address mod 8): additive, so the kernel
// that writes the bytes can take it on the way out instead of a second pass reading 33 MB per C2 frame back (C5: 53 -> 43 us per frame).  Every lane adds the
// elements it stores, shifted to their place in the word, into an LDS slot of its own (one ds_add_u64, no return); at a frame change and at the end a wave folds
// its 64 slots together and writes ONE word of the launch's table of partial sums (GfwYuvArgs.ck_part), which gfw_ck_finish adds up.  Builds without the option carry none of this.
#ifndef GFW_LDS_ADD
#define GFW_LDS_ADD(p, v) ((void)__hip_atomic_fetch_add((p), (v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT))      // ds_add_u64: nothing comes back, nothing to wait for
#endif
#if GFW_BAKE && GFW_BK_checksum
#define GFW_CK 1
__shared__ unsigned long long gfw_ck_slot[256];       // a lane's stores outside the branch-free lane-row
__shared__ unsigned long long gfw_ck_wave[4];         // a wave's sum while it is being folded
#else
#define GFW_CK 0
#endif
#define GFW_CK_ELSEWHERE ((unsigned long long *)8)
template <typename T>
__device__ __forceinline__ void gfw_ck(uint32_t off, T v, unsigned long long *acc = nullptr) {
#if GFW_CK
    static_assert(sizeof(T) <= 4, "one element of a plane");
    unsigned long long bits;
    if constexpr (sizeof(T) == 1) bits = __builtin_bit_cast(uint8_t, v);
    else if constexpr (sizeof(T) == 2) bits = __builtin_bit_cast(uint16_t, v);
    else bits = __builtin_bit_cast(uint32_t, v);
    // the element's place in its word: from the offset alone — the host sends only frames whose planes start on a word (8-byte) boundary here, and whose planes and
    // strides are element-aligned, so that an element never straddles a word
    if (acc == GFW_CK_ELSEWHERE) return;            // (the caller accounts for this store itself)
    const unsigned sh = (off & 7u) * 8u;
    if (acc) *acc += bits << sh;                    // the branch-free lane-row: a register pair of the lane (folded into the slot before the wave's fold)
    else {
        unsigned slot = threadIdx.y * 64 + threadIdx.x;
        asm("" : "+v"(slot));                       // opaque: the slot's address is worked out at the (rare) store, not kept — in scratch — across the kernel
        GFW_LDS_ADD(&gfw_ck_slot[slot], bits << sh);
    }
#else
    (void)off; (void)v; (void)acc;
#endif
}
// two adjacent elements that left as one store: one addition when the pair sits inside a word (its offset a multiple of its size), else element by element
// (`inside`: the caller knows it does — a pair at an even pixel of a plane whose stride is a multiple of the pair, a literal in a baked build: no test per store)
template <typename E>
__device__ __forceinline__ void gfw_ck_pair(uint32_t off, uint32_t v0, uint32_t v1, unsigned long long *acc = nullptr, bool inside = false) {
#if GFW_CK
    if constexpr (sizeof(E) == 4) { gfw_ck<uint32_t>(off, v0, acc); gfw_ck<uint32_t>(off + 4u, v1, acc); }
    else if (inside || (off & (2u * (unsigned)sizeof(E) - 1u)) == 0u) {
        if constexpr (sizeof(E) == 1) gfw_ck<uint16_t>(off, (uint16_t)((v0 & 0xffu) | (v1 << 8)), acc);
        else gfw_ck<uint32_t>(off, (v0 & 0xffffu) | (v1 << 16), acc);
    } else { gfw_ck<E>(off, (E)v0, acc); gfw_ck<E>(off + (unsigned)sizeof(E), (E)v1, acc); }
#else
    (void)off; (void)v0; (void)v1; (void)acc; (void)inside;
#endif
}
namespace {

// Wave votes as one compare into a scalar pair and one scalar compare with EXEC: the library's __all / __any go through a 0/1 select and a second
// compare (two more vector instructions and a VCC hazard per vote; the branch-free row votes three times per pixel pair).
#if defined(GFW_HOST_INTERPRETER)
#define gfw_all(p) __all(p)
#define gfw_any(p) __any(p)
static inline float gfw_uniform(float v) { return v; }
#else
// a wave-uniform value the compiler computed with vector instructions: into a scalar register (else it occupies a VGPR of every lane for the whole kernel)
__device__ __forceinline__ float gfw_uniform(float v) { return __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, v))); }
__device__ __forceinline__ bool gfw_all(bool p) { return __builtin_amdgcn_ballot_w64(p) == __builtin_amdgcn_ballot_w64(true); }
__device__ __forceinline__ bool gfw_any(bool p) { return __builtin_amdgcn_ballot_w64(p) != 0ull; }
#endif
// A vote over a conjunction, assembled from the ballots of its terms (each a compare straight into a scalar pair; the conjunction itself would be
// rebuilt as a 0/1 select and compared again): `v &= gfw_lanes(test)` per term, then gfw_all_lanes(v).
#if defined(GFW_HOST_INTERPRETER)
typedef bool GfwVote;
#define GFW_VOTE_ALL true
static inline bool gfw_lanes(bool p) { return p; }
static inline bool gfw_all_lanes(bool v) { return __all(v); }
static inline bool gfw_vote_select(bool cur, bool cond, bool val) { return cond ? val : cur; }
static inline bool gfw_vote_lane(bool v, int) { return v; }
#define gfw_vote_failed(v) __ballot(!(v))
#else
typedef unsigned long long GfwVote;
#define GFW_VOTE_ALL (~0ull)
#define gfw_lanes(p) __builtin_amdgcn_ballot_w64(p)
__device__ __forceinline__ bool gfw_all_lanes(GfwVote v) { const GfwVote e = __builtin_amdgcn_ballot_w64(true); return (v & e) == e; }
__device__ __forceinline__ GfwVote gfw_vote_select(GfwVote cur, GfwVote cond, GfwVote val) { return (cur & ~cond) | (cond & val); }     // per lane: cond ? val : cur
__device__ __forceinline__ bool gfw_vote_lane(GfwVote v, int lane) { return ((v >> (unsigned)lane) & 1ull) != 0ull; }
__device__ __forceinline__ unsigned long long gfw_vote_failed(GfwVote v) { return ~v & __builtin_amdgcn_ballot_w64(true); }       // the active lanes whose predicate was false
#endif

// 32-phase bicubic / Lanczos4 tap table (one constant copy per translation unit)
__device__
#include "gfw_coeffs.inc"


struct IeeeOps {
    static __device__ __forceinline__ void div2(float a1, float a2, float b, float &q1, float &q2) { q1 = a1 / b; q2 = a2 / b; }
    static __device__ __forceinline__ float div(float a, float b) { return a / b; }
    static __device__ __forceinline__ float sqrt(float x) { return sqrtf(x); }
    static __device__ __forceinline__ float atan_pos(float x) { return gfw_atanf(x); }
};
struct LeanOps {
    static __device__ __forceinline__ void div2(float a1, float a2, float b, float &q1, float &q2) {
        const GfwRcp d = gfw_rcp_prepare(b);
        q1 = gfw_div_prepared(a1, d); q2 = gfw_div_prepared(a2, d);
    }
    static __device__ __forceinline__ float div(float a, float b) { return gfw_div_lean(a, b); }
    static __device__ __forceinline__ float sqrt(float x) {
        if (__builtin_expect(x < 8.271806125530277e-25f, 0)) return sqrtf(x);    // below 2^-80 (zero included: the optical centre): generic path
        return gfw_sqrt_lean(x);
    }
    // glibc atanf, wave-specialised: when every active lane is below 0.4375 the reduction (and its division)
    // disappears; otherwise the select-based single-division form (gfw_fastmath.h).
    static __device__ __forceinline__ float atan_pos(float x) {
        if (__all(x < 0.4375f)) {
            const float aT0 = 3.3333334327e-01f, aT1 = -2.0000000298e-01f, aT2 = 1.4285714924e-01f, aT3 = -1.1111110449e-01f,
                        aT4 = 9.0908870101e-02f, aT5 = -7.6918758452e-02f, aT6 = 6.6610731184e-02f, aT7 = -5.8335702866e-02f,
                        aT8 = 4.9768779427e-02f, aT9 = -3.6531571299e-02f, aT10 = 1.6285819933e-02f;
            const float z = x * x, w = z * z;
            const float s1 = z * (aT0 + w * (aT2 + w * (aT4 + w * (aT6 + w * (aT8 + w * aT10)))));
            const float s2 = w * (aT1 + w * (aT3 + w * (aT5 + w * (aT7 + w * aT9))));
            return x - x * (s1 + s2);
        }
        return gfw_atanf_pos_tab(x);
    }
};

// Lens + per-plane uniforms, VGPR-resident for the pixel loops.
struct Lens {
    float f0, f1, c0, c1, k0, k1, k2, k3, t2x, t2y, rl2;
};
struct Maps {                       // source_rect maps: u * mul / den  (den, rcp shared by luma and chroma)
    float mul_lx, mul_ly, mul_cx, mul_cy, den_x, rcp_x, den_y, rcp_y;
};
// INF_SAFE: an infinite coordinate must stay infinite (x*mul/den in IEEE; the reference then casts it to i32::MIN/MAX and
// reads background), but the remainder step turns it into inf - inf = NaN; clamping the NaN remainder to a finite value
// restores q0's infinity and changes nothing for finite or NaN inputs.  The specialised fisheye projection cannot produce
// an infinite coordinate (a*s is bounded by theta_d), so only the generic-model instantiation pays for the guard.
template <bool INF_SAFE>
__device__ __forceinline__ float map_c(float x, float mul, float den, float rcp) {
    const float a = x * mul;
    const float q0 = a * rcp;
    float r0 = __builtin_fmaf(-den, q0, a);
    if (INF_SAFE) r0 = fmaxf(r0, -3.4028234664e38f);
    return __builtin_fmaf(r0, rcp, q0);
}

// The chroma plane's source_rect map of a coordinate x whose luma map l = map_c(x, mul_l, den, rcp) is already known (baked builds: the
// multipliers are literals, the branches fold).  Same multiplier (4:2:2's rows, 4:4:4): the same operations, l itself.  Half the
// multiplier (a subsampled axis): every operation of map_c scales exactly by 1/2 — power-of-two scaling commutes with round-to-nearest
// while nothing underflows — so the result is l/2; coordinates below 2^-100 (where the residual of the division could go subnormal) land in
// the same bins either way (below).
template <bool INF_SAFE>
__device__ __forceinline__ float chroma_from_luma(float l, float x, float mul_c, float mul_l, float den, float rcp) {
    if (mul_c == mul_l) return l;
    if (2.0f * mul_c == mul_l) {
        // Every consumer of a chroma coordinate bins it first: round((c - OFFSET) * 32), OFFSET 0 / 1 / 3 (raw_bin, make_bins2).  Where the halving is exact — all but
        // |l| < 2^-100, whose residual could go subnormal — l/2 IS the map's value.  Below 2^-100 both l/2 and the map's own value are below 2^-100 in magnitude, and
        // every bin computed from such a coordinate is the same: 32 c rounds to 0, c - OFFSET is -OFFSET exactly.  (Round 4 sent the whole wave through the full
        // evaluation on a tiny lane: a compare, a select and the map itself, twelve instructions per lane-row for a case that changes nothing.)
        return 0.5f * l;
    }
    return map_c<INF_SAFE>(x, mul_c, den, rcp);
}

// opencv_fisheye.rs:72-95 on (X/W, Y/W); then *f, +c (cpu_undistort.rs:155,167)
template <class Ops>
__device__ __forceinline__ void fisheye_project(float X, float Y, float W, const Lens &L, bool k_all_zero, float &u, float &v) {
    float a, b;
    Ops::div2(X, Y, W, a, b);
    if (!k_all_zero) {
        const float r = Ops::sqrt(a * a + b * b);
        const float t = Ops::atan_pos(r);
        const float t2 = t * t, t4 = t2 * t2, t6 = t4 * t2, t8 = t4 * t4;
        const float td = t * (1.0f + L.k0 * t2 + L.k1 * t4 + L.k2 * t6 + L.k3 * t8);
        const float s = (r == 0.0f) ? 1.0f : Ops::div(td, r);
        a = a * s; b = b * s;
    }
    u = a * L.f0 + L.c0;
    v = b * L.f1 + L.c1;
}

// rotate_and_distort (cpu_undistort.rs:133-228) restricted to the eligible configuration
// (no mesh, translation3d == 0; IBIS terms, digital lens and refraction only through the generic-model instantiation).  ma/mb/m8 = the 9 matrix entries of the chosen row.
template <int MODEL>
__device__ __forceinline__ GfwPt rd(float px, float py, const float4 ma, const float4 mb, const float m8, const float *ext, const Lens &L, const GfwYuvArgs &A) {
    const float X = (px * ma.x) + (py * ma.y) + ma.z;
    const float Y = (px * ma.w) + (py * mb.x) + mb.y;
    const float W = (px * mb.z) + (py * mb.w) + m8;
    GfwPt o{0.0f, 0.0f, false};
    if (MODEL == GFW_MODEL_OPENCV_FISHEYE) {
        // proven operand range of the lean divide: |X|,|Y| <= 2^19, W in [2^-20, 2^20]  (=> |a|,|b| <= 2^39).  The range test subsumes the
        // reference's `w > 0` (:137), so the usual pixel pays one test; everything else — non-positive or tiny W, huge operands, NaN —
        // takes the side branch with the reference's own order of tests and the generic IEEE expansions.
        const float mag = fmaxf(fmaxf(fabsf(X), fabsf(Y)), W);
        const bool lean = (mag <= 524288.0f) && (W >= 9.5367431640625e-07f);
        if (__builtin_expect(lean, 1)) {
            if (L.rl2 > 0.0f && (X * X + Y * Y) > L.rl2 * W) return o;
            o.ok = true;
            fisheye_project<LeanOps>(X, Y, W, L, AF(k_all_zero) != 0, o.x, o.y);
        } else {
            if (!(W > 0.0f)) return o;
            if (L.rl2 > 0.0f && (X * X + Y * Y) > L.rl2 * W) return o;
            o.ok = true;
            fisheye_project<IeeeOps>(X, Y, W, L, AF(k_all_zero) != 0, o.x, o.y);      // generic IEEE expansions
        }
#if GFW_BAKE
        // a digital lens on top of the fisheye (GoPro SuperView / HyperView clips, flags & 2: cpu_undistort.rs:216-220) in a baked build:
        // the lens is a literal, so the specialised projection above serves these clips too (ahead of time they take the generic-model
        // instantiation); same position in the chain — after `+ c` — and the reference's own arithmetic (gfw_warp.h)
        if (AF(extras) & 2) {
            float d0, d1;
            gfw_lens::distort<GFW_BK_digital>(GFW_BK_digital, o.x, o.y, 1.0f, A.kp, A.common, d0, d1);
            o.x = d0; o.y = d1;
        }
#endif
    } else {
        if (!(W > 0.0f)) return o;                                                   // :137
        if (L.rl2 > 0.0f && (X * X + Y * Y) > L.rl2 * W) return o;                   // :139
        o.ok = true;
        // every lens model through the generic IEEE routines, plus the optional stages of rotate_and_distort in the
        // reference's order: refraction (:143-152), model, *f, IBIS/OIS rotate + shift (:157-165), +c, digital lens (:216-220)
        float Wd = W;
        if ((AF(extras) & 4) && W != 0.0f) {
            const float r = sqrtf(X * X + Y * Y) / W;
            const float sin_theta_d = (r / sqrtf(1.0f + r * r)) * A.kp.light_refraction_coefficient;
            const float r_d = sin_theta_d / sqrtf(1.0f - sin_theta_d * sin_theta_d);
            if (r_d != 0.0f) Wd *= r / r_d;
        }
        float du, dv;
        gfw_lens::distort<(MODEL == GFW_MODEL_GENERIC_EXTRA ? -1 : MODEL)>(AF(model), X, Y, Wd, A.kp, A.common, du, dv);
        float u = du * L.f0, v = dv * L.f1;
        if (AF(extras) & 1) {
            const float m9 = ext[1], m10 = ext[2], m11 = ext[3], m12 = ext[4], m13 = ext[5];
            if (m9 != 0.0f || m10 != 0.0f || m11 != 0.0f || m12 != 0.0f || m13 != 0.0f) {
                const float cos_a = ext[6], sin_a = ext[7];               // cosf(-m11), sinf(-m11) from the host libm
                const float nu = cos_a * u - sin_a * v - m9 + m12;
                const float nv = sin_a * u + cos_a * v - m10 + m13;
                u = nu; v = nv;
            }
        }
        u = u + L.c0; v = v + L.c1;
        if (MODEL == GFW_MODEL_GENERIC_EXTRA && (AF(extras) & 32)) gfw_mesh_apply(u, v, A.kp, A.common);   // Sony mesh + focal-plane distortion (:169-214)
        if (AF(extras) & 2) {
            float d0, d1;
            gfw_lens::distort<GFW_BAKED_DIGITAL_MODEL>(GFW_CLIP_DIGITAL, u, v, 1.0f, A.kp, A.common, d0, d1);
            u = d0; v = d1;
        }
        o.x = u; o.y = v;
    }
    // input_{horizontal,vertical}_stretch (cpu_undistort.rs:222-223; anamorphic lens profiles): <= 0.001 is skipped and x / 1.0 == x, so the host raises the
    // flags only for a real divisor — then an IEEE division, as the reference's
    if (AF(hstretch_div)) o.x = o.x / AF(hstretch);
    if (AF(vstretch_div)) o.y = o.y / AF(vstretch);
    return o;
}
// The specialised fisheye projection WITHOUT a branch (round 4).  rd<>'s lean / generic split, the square root's tiny-operand case and the r == 0
// case are each a divergent `if` in the pixel loop, i.e. s_and_saveexec / s_cbranch_execz / s_or exec scaffolding around code that practically
// every lane runs: scalar instructions take the same issue slots as vector ones on MI355X (tools/microbench_mix.hip: one SALU per VALU doubles a
// loop's time; an always-taken per-lane `if` costs ~9.5 SIMD-cycles), and the kernel carried 0.27 of them per VALU instruction.  Here every lane
// evaluates the lean sequence — the very operations of rd<>'s lean branch, in its order — and reports in `rare` whether its operands were outside
// what that sequence is proven for (the range test, rr < 2^-80 which includes the optical centre's r == 0, the r-limit); the caller asks the WAVE
// once (`__any(rare)`: a uniform branch, no exec bookkeeping) and sends exactly those lanes through rd<> itself.  A rare lane's values here are
// garbage by design (possibly NaN / inf) and are never used.
template <int NP>
__device__ __forceinline__ void rd_lean_nobranch(const float *px, const float *py, const float4 *ma, const float4 *mb, const float *m8, const Lens &L, const GfwYuvArgs &A,
                                                 float *u, float *v, bool *rare) {
    // the NP pixels of a lane side by side, stage by stage: one basic block, so the two dependency chains interleave (the VCC / packed-result hazards of
    // one pixel are filled with the other's instructions instead of s_nop) and the arctangent's small-angle question is put to the wave once for all of them
    float a[NP], b[NP], r[NP];
    #pragma unroll
    for (int i = 0; i < NP; ++i) {
        const float X = (px[i] * ma[i].x) + (py[i] * ma[i].y) + ma[i].z;
        const float Y = (px[i] * ma[i].w) + (py[i] * mb[i].x) + mb[i].y;
        const float W = (px[i] * mb[i].z) + (py[i] * mb[i].w) + m8[i];
        const float mag = fmaxf(fmaxf(fabsf(X), fabsf(Y)), W);
        bool odd = !((mag <= 524288.0f) && (W >= 9.5367431640625e-07f));           // NaN operands fail the tests and are odd
        if (L.rl2 > 0.0f) odd = odd | ((X * X + Y * Y) > L.rl2 * W);             // :139 — the side path answers "no point" with the reference's own test
        LeanOps::div2(X, Y, W, a[i], b[i]);
        rare[i] = odd;
    }
    if (!(AF(k_all_zero) != 0)) {
        GfwVote small = GFW_VOTE_ALL;
        #pragma unroll
        for (int i = 0; i < NP; ++i) {
            const float rr = a[i] * a[i] + b[i] * b[i];
            // rr in [2^-80, 2^50): below, the generic sqrt and the optical centre's s = 1; above, atanf's r >= 2^25 record — one unsigned range test on the bit pattern (NaN fails it)
            rare[i] = rare[i] | !((gfw_f2u(rr) - 0x17800000u) < (0x58800000u - 0x17800000u));
            r[i] = gfw_sqrt_lean(rr);
            small = small & gfw_lanes(r[i] < 0.4375f);
        }
        float t[NP];
        if (gfw_all_lanes(small)) {
            const float aT0 = 3.3333334327e-01f, aT1 = -2.0000000298e-01f, aT2 = 1.4285714924e-01f, aT3 = -1.1111110449e-01f,
                        aT4 = 9.0908870101e-02f, aT5 = -7.6918758452e-02f, aT6 = 6.6610731184e-02f, aT7 = -5.8335702866e-02f,
                        aT8 = 4.9768779427e-02f, aT9 = -3.6531571299e-02f, aT10 = 1.6285819933e-02f;
            #pragma unroll
            for (int i = 0; i < NP; ++i) {
                const float x = r[i], z = x * x, w = z * z;
                const float s1 = z * (aT0 + w * (aT2 + w * (aT4 + w * (aT6 + w * (aT8 + w * aT10)))));
                const float s2 = w * (aT1 + w * (aT3 + w * (aT5 + w * (aT7 + w * aT9))));
                t[i] = x - x * (s1 + s2);
            }
        } else {
            #pragma unroll
            for (int i = 0; i < NP; ++i) t[i] = gfw_atanf_pos_key(r[i]);
        }
        #pragma unroll
        for (int i = 0; i < NP; ++i) {
            const float t2 = t[i] * t[i], t4 = t2 * t2, t6 = t4 * t2, t8 = t4 * t4;
            const float td = t[i] * (1.0f + L.k0 * t2 + L.k1 * t4 + L.k2 * t6 + L.k3 * t8);
            const float s = LeanOps::div(td, r[i]);
            a[i] = a[i] * s; b[i] = b[i] * s;
        }
    }
    #pragma unroll
    for (int i = 0; i < NP; ++i) { u[i] = a[i] * L.f0 + L.c0; v[i] = b[i] * L.f1 + L.c1; }
}

template <int MODEL>
__device__ __forceinline__ GfwPt rd_row(float px, float py, int idx, const float *matrices, const Lens &L, const GfwYuvArgs &A) {
    const float *m = reinterpret_cast<const float *>(reinterpret_cast<const uint8_t *>(matrices) + (uint32_t)idx * (uint32_t)(GFW_MAT_STRIDE * sizeof(float)));   // idx >= 0
    return rd<MODEL>(px, py, *reinterpret_cast<const float4 *>(m), *reinterpret_cast<const float4 *>(m + 4), m[8], m + 8, L, A);
}

// f32::round (half away from zero) then `as i32`: rndne is exact except on ties, which take the side branch.
// `x.round() as i32` (half away from zero, then truncating saturating cast) without the tie branch:
// trunc(x + copysign(pred(0.5), x)) — equal to the cast of roundf(x) for every one of the 2^32 floats
// (tests/test_math_host.py checks this exhaustively); the cast itself truncates.
__device__ __forceinline__ int round_i32(float x) {
    return gfw_f2i(x + copysignf(0x1.fffffep-2f, x));
}
// round_i32(u * 32): the product is exact (a power of two; coordinates this large or small — beyond 2^122, below 2^-120 — are not finite pixel positions and land in
// the same saturated / zero bin either way), so the multiply and the add are ONE fused operation with the same single rounding (round 5: an instruction per coordinate)
__device__ __forceinline__ int round32_i32(float u) {
    return gfw_f2i(__builtin_fmaf(u, 32.0f, copysignf(0x1.fffffep-2f, u)));
}
// f32::min(v, limit) with the hardware's IEEE-mode v_min_f32 (non-NaN operand wins, as Rust's does): spares the
// canonicalising v_max the compiler puts in front of fminf for a uniform operand.
__device__ __forceinline__ float min_limit(float v, float limit) {
    float r; asm("v_min_f32 %0, %1, %2" : "=v"(r) : "v"(v), "s"(limit)); return r;
}

// Byte offset of a row: 24-bit multiply (v_mul_i32_i24 / v_mad_i32_i24, full rate) — the 32-bit v_mul_lo_u32 is a quarter-rate instruction and
// sat in every sample's address.  Row indices and pitches of the planes this kernel serves are below 2^23 (host-checked); a sample far
// outside the plane may wrap here, but such a sample never passes the interior test that guards the use of the offset.
__device__ __forceinline__ int row_off(int y, int stride) {
    int r;                                     // spelled out: the compiler turns the mul24 intrinsic back into a 32-bit multiply where it cannot prove the range
    asm("v_mul_i32_i24 %0, %1, %2" : "=v"(r) : "v"(y), "s"(stride));
    return r;
}
// `as u32` of a float known to be finite or NaN: v_cvt_u32_f32 (truncate; negative and NaN -> 0; saturating)
__device__ __forceinline__ uint32_t gfw_f2u_trunc(float v) { uint32_t r; asm("v_cvt_u32_f32 %0, %1" : "=v"(r) : "v"(v)); return r; }

// ---- LUT taps (cpu_undistort.rs:371-418): I = 2 bilinear, 4 bicubic, 8 Lanczos4 --------------------------------
// The I x-weights and I y-weights of a sample stay in the LDS copy of the table (32 phases x I floats per filter); a
// Bins holds the two row pointers.  (Keeping 2*I weights per sample in registers cost the bicubic / Lanczos4 kernels
// their occupancy: 148 VGPRs for I = 8.)
template <int I> struct Bins { int sx, sy; const float *tx, *ty; };
template <int I> __device__ __forceinline__ int raw_bin(float u) {           // round((u - offset) * 32): the sample's column / row and its 1/32 phase (:374, :380-381)
    constexpr float OFFSET = (I == 2) ? 0.0f : (I == 4 ? 1.0f : 3.0f);
    return round32_i32(u - OFFSET);
}
// The bins of a source coordinate of plane P.  The kernel's planes begin at the first pixel of their source rect (the host rebases the pointers); the reference maps
// a coordinate INTO the rect — map_coord's `+ out_min`, cpu_undistort.rs:510-515: `u * srw / W + srx`, one more rounding — and bins that sum.  So the origin is added
// as the reference adds it and 32 * origin comes off the bins again (exact integer arithmetic).  A buffer without rects has origin 0: ox32 / oy32 are literals of a
// baked build, a uniform branch ahead of time, and the coordinate is binned as before (u + (+0) differs from u only for u = -0, which bins to 0 either way).
template <int I> __device__ __forceinline__ int bin_x(float u, const GfwYuvPlane &P) { return P.ox32 ? raw_bin<I>(u + P.org_x) - P.ox32 : raw_bin<I>(u); }
template <int I> __device__ __forceinline__ int bin_y(float v, const GfwYuvPlane &P) { return P.oy32 ? raw_bin<I>(v + P.org_y) - P.oy32 : raw_bin<I>(v); }
template <int I>
__device__ __forceinline__ Bins<I> bins_of(int sx0, int sy0, const float *lut) {
    Bins<I> b;
    b.sx = sx0 >> 5; b.sy = sy0 >> 5;
    constexpr int IND = (I == 2) ? 0 : (I == 4) ? 64 : 192, SHIFT = (I >> 2) + 1;       // :373-375
    b.tx = lut + IND + ((sx0 & 31) << SHIFT); b.ty = lut + IND + ((sy0 & 31) << SHIFT);
    return b;
}
template <int I>
__device__ __forceinline__ Bins<I> make_bins(float u, float v, const float *lut) { return bins_of<I>(raw_bin<I>(u), raw_bin<I>(v), lut); }
// float samples — f32, and since round 5 the f16 of packed RGBAf16 (pixel_formats.rs:227-246: half::f16 to_f32 on load, from_f32 = IEEE round-to-nearest-even on store:
// the hardware's v_cvt_f32_f16 / v_cvt_f16_f32): every tap is converted to f32 first, so the float arithmetic of a sample (leading zero adds kept: signed zeros, no
// saturation) is the same for both; only the load and the store know the width.
template <typename T> struct is_f32 { static constexpr bool value = false; };
template <> struct is_f32<float> { static constexpr bool value = true; };
template <> struct is_f32<_Float16> { static constexpr bool value = true; };
// Taps that straddle the source rect: out-of-rect taps read `bg`, out-of-rect rows contribute bg*cy
// (cpu_undistort.rs:391-411), in the reference's exact operation order.
template <typename T, int N, int I>
__device__ __forceinline__ void taps_edge(const uint8_t *src, int stride, const Bins<I> &b, int w, int h, const float *bg, float limit, float *out) {
    float sum[N];
    #pragma unroll
    for (int c = 0; c < N; ++c) sum[c] = 0.0f;
    #pragma unroll 1        // the rare path: rolled, weights read from the table as they are needed
    for (int yp = 0; yp < I; ++yp) {
        const int yy = b.sy + yp;
        const float wy = b.ty[yp];
        if (yy >= 0 && yy < h) {
            const T *row = reinterpret_cast<const T *>(src + (int64_t)yy * stride);
            float xs[N];
            #pragma unroll
            for (int c = 0; c < N; ++c) xs[c] = 0.0f;
            #pragma unroll 1
            for (int xp = 0; xp < I; ++xp) {
                const int xx = b.sx + xp;
                const bool in = xx >= 0 && xx < w;
                const float wx = b.tx[xp];
                #pragma unroll
                for (int c = 0; c < N; ++c) { const float px = in ? (float)row[(int64_t)xx * N + c] : bg[c]; xs[c] = xs[c] + px * wx; }
            }
            #pragma unroll
            for (int c = 0; c < N; ++c) sum[c] = sum[c] + xs[c] * wy;
        } else {
            #pragma unroll
            for (int c = 0; c < N; ++c) sum[c] = sum[c] + bg[c] * wy;
        }
    }
    #pragma unroll
    for (int c = 0; c < N; ++c) out[c] = fminf(sum[c], limit);
}
// All I x I taps inside.  Bilinear on integer pixels: every tap is >= +0, so the reference's leading zero-adds
// (xsum = 0 + p*c, sum = 0 + xs*cy) are exact identities and are dropped; everywhere else (negative weights, f32
// pixels with -0 / negative values) they are kept so that signed zeros come out as the reference's.
template <typename T, int N, int I>
__device__ __forceinline__ void taps_inside(const uint8_t *src, int off0, int stride, const Bins<I> &b, float limit, float *out) {
    float cx[I];
    #pragma unroll
    for (int i = 0; i < I; ++i) cx[i] = b.tx[i];
    if (N == 1 && !is_f32<T>::value && I > 2) {
        // single-channel integer planes (Y, U, V): the I taps of a row are I*sizeof(T) contiguous bytes.  They are fetched as
        // ALIGNED dwords and funnel-shifted into place: a row fetch whose address is not 4-byte aligned (every odd u16
        // pixel) costs the texture-address unit 65 cycles instead of 18 (profiles/r01_membench_tap_row_fetch.txt), and that
        // was what the Lanczos4 kernel waited for.
        // Round 4 (the row body was 45 issue slots, 7 of them scalar): ND + 1 dwords are fetched UNCONDITIONALLY — the interior test keeps
        // tap_margin pixels clear of the row end, so the last dword never leaves the row (it used to be read under a per-lane branch when the
        // sample was misaligned) — addresses are 32-bit lane offsets on the uniform plane base (plane bases are 4-byte aligned: host-checked),
        // and for integer pixels the reference's leading zero-adds (xs = 0 + p*c, sum = 0 + xs*cy) are dropped: they only decide the sign of a
        // zero, which no later operation of an integer sample can see (`as u8 / u16` maps both zeros to 0).
        constexpr int ND = (I * (int)sizeof(T)) / 4;
        // one tap row from its ND + 1 fetched dwords: funnel-shifted into place, converted, multiplied, added in the reference's order
        auto row_sum = [&](const uint32_t *w, unsigned sh) -> float {
            float xs = 0.0f;
            #pragma unroll
            for (int j = 0; j < ND; ++j) {
                const uint32_t d = __builtin_amdgcn_alignbit(w[j + 1], w[j], sh);       // ({w[j+1], w[j]} >> sh)[31:0]
                if (sizeof(T) == 2) {
                    const float t0 = (float)(d & 0xffffu) * cx[2 * j];
                    xs = (j == 0) ? t0 : xs + t0;
                    xs = xs + (float)(d >> 16) * cx[2 * j + 1];
                } else {
                    const float t0 = (float)(d & 0xffu) * cx[4 * j];
                    xs = (j == 0) ? t0 : xs + t0;
                    xs = xs + (float)((d >> 8) & 0xffu) * cx[4 * j + 1];
                    xs = xs + (float)((d >> 16) & 0xffu) * cx[4 * j + 2];
                    xs = xs + (float)(d >> 24) * cx[4 * j + 3];
                }
            }
            return xs;
        };
        auto fetch = [&](const uint32_t *wp, bool extra, uint32_t *w) {
            #pragma unroll
            for (int j = 0; j < ND; ++j) w[j] = wp[j];
            w[ND] = extra ? wp[ND] : 0u;
        };
        float s1 = 0.0f;
        if ((stride & 3) == 0) {
            // the usual case (row pitch a multiple of 4 bytes): the misalignment is the same for every tap row of the sample.  The rows go in groups of R: all
            // fetches of a group first, then its arithmetic — said in so many words because the ROCm 7.2 compiler, left to unroll a fetch-and-convert loop, issued
            // `load, wait, load, wait` for one of the two luma samples where the 7.0 one clustered the loads (profiles/r04_ab_lut_rows.txt)
            constexpr int R = GFW_TAP_ROW_UNROLL(I, T) < I ? GFW_TAP_ROW_UNROLL(I, T) : I;
            static_assert(I % R == 0, "tap rows per group");
            const unsigned mis = (unsigned)off0 & 3u, sh = mis * 8u;
            // 16-bit Lanczos4 is bound by the fetches themselves (dwordx4 + dword per row: the second one only for the misaligned half of the samples —
            // unconditional it measured 170 against 157 us per C2 frame); everywhere else the branch costs more than the fetch it saves
            // (bicubic 81 -> 67 us, 8-bit Lanczos4 141 -> 106)
            constexpr bool COND = sizeof(T) == 2 && I == 8;      // the last dword only for the misaligned half of the samples
            uint32_t aoff = (uint32_t)off0 & ~3u;
            #pragma unroll 1
            for (int y0 = 0; y0 < I; y0 += R) {
                uint32_t w[R][ND + 1];
                #pragma unroll
                for (int r = 0; r < R; ++r) {
                    const uint32_t *wp = reinterpret_cast<const uint32_t *>(src + (uint32_t)(aoff + (uint32_t)(r * stride)));     // base + ONE zero-extended 32-bit lane offset (saddr form)
                    #pragma unroll
                    for (int j = 0; j < ND + (COND ? 0 : 1); ++j) w[r][j] = wp[j];       // (ND + 1 adjacent dwords: one wider fetch)
                    if (COND) w[r][ND] = 0u;
                }
                if (COND && mis != 0u) {                 // ONE per-lane region for the group's last dwords (it was one per row: three more sets of exec bookkeeping)
                    #pragma unroll
                    for (int r = 0; r < R; ++r) w[r][ND] = reinterpret_cast<const uint32_t *>(src + (uint32_t)(aoff + (uint32_t)(r * stride) + 4u * ND))[0];
                }
                #pragma unroll
                for (int r = 0; r < R; ++r) s1 = s1 + row_sum(w[r], sh) * b.ty[y0 + r];      // (the first of these adds is the reference's 0 + xs*cy: kept)
                aoff += (uint32_t)(R * stride);
            }
        } else {
            #pragma unroll 1
            for (int yp = 0; yp < I; ++yp) {
                const uint32_t o = (uint32_t)(off0 + yp * stride);
                const unsigned mis = o & 3u;
                uint32_t w[ND + 1];
                fetch(reinterpret_cast<const uint32_t *>(src + (o & ~3u)), mis != 0, w);
                s1 = s1 + row_sum(w, mis * 8u) * b.ty[yp];
            }
        }
        out[0] = fminf(s1, limit);
        return;
    }
    float sum[N];
    #pragma unroll
    for (int c = 0; c < N; ++c) sum[c] = 0.0f;
    // multi-channel and float pixels: the rows in groups too — a group's fetches first, then its arithmetic (left to one unrolled loop the shipped compiler issued
    // `fetch, wait` row by row for the interleaved chroma of NV12 / P010: tests/test_kernel_fetch_clusters.py).  Integer pixels only: 16-byte float4 taps would hold I * 4 registers per row.
    constexpr int RG = is_f32<T>::value ? 1 : (I * N * (int)sizeof(T) <= 16 ? (I < 4 ? I : 4) : 2);
    #pragma unroll
    for (int y0 = 0; y0 < I; y0 += RG) {
        T v[RG][I * N];
        #pragma unroll
        for (int r = 0; r < RG; ++r) {
            const T *row = reinterpret_cast<const T *>(src + (uint32_t)(off0 + (y0 + r) * stride));
            #pragma unroll
            for (int e = 0; e < I * N; ++e) v[r][e] = row[e];
        }
        #pragma unroll
        for (int r = 0; r < RG; ++r) {
            float xs[N];
            #pragma unroll
            for (int c = 0; c < N; ++c) xs[c] = 0.0f;
            #pragma unroll
            for (int xp = 0; xp < I; ++xp) {
                #pragma unroll
                for (int c = 0; c < N; ++c) xs[c] = xs[c] + (float)v[r][xp * N + c] * cx[xp];
            }
            #pragma unroll
            for (int c = 0; c < N; ++c) sum[c] = sum[c] + xs[c] * b.ty[y0 + r];
        }
    }
    #pragma unroll
    for (int c = 0; c < N; ++c) out[c] = fminf(sum[c], limit);
}
// tap_margin: pixels kept clear of the row end by the aligned dword fetch of taps_inside, which reads ND + 1 dwords from the aligned address at or below the
// first tap: up to 4/sizeof(T) pixels past the last tap when the sample IS aligned.  Samples closer to the edge take the exact edge path.
template <typename T, int N, int I> __device__ constexpr int tap_margin() { return (N == 1 && !is_f32<T>::value && I > 2) ? 4 / (int)sizeof(T) : 0; }
template <typename T, int N, int I>
__device__ __forceinline__ bool bins_inside(const Bins<I> &b, int w, int h) {
    constexpr int TAP_MARGIN = tap_margin<T, N, I>();
    return w >= I + TAP_MARGIN && h >= I && (unsigned)b.sx <= (unsigned)(w - I - TAP_MARGIN) && (unsigned)b.sy <= (unsigned)(h - I);
}
// Does the saturating `as u8/u16` cast need its upper clamp?  Every value a sample can take is min(sum, limit) or bg[c]; when both are
// clip constants no larger than the type's maximum the truncating conversion alone is the cast (negative and NaN -> 0 in hardware).
// Decidable at compile time in a baked build only; elsewhere the clamp stays.
template <typename T>
__device__ __forceinline__ bool px_needs_sat(const float *bg, int n, float limit) {
#if GFW_BAKE
    const float top = sizeof(T) == 1 ? 255.0f : 65535.0f;
    bool fits = limit <= top;
    for (int c = 0; c < n; ++c) fits = fits && bg[c] <= top;
    return !fits;
#else
    (void)bg; (void)n; (void)limit;
    return true;
#endif
}
// remap_colorrange (cpu_undistort.rs:254-260; applied to the finished pixel — sample or background — right before the cast, :619-621): every lane times the plane's
// scale, then 16 on lanes 0 and 1.  fix: 0 off, 1 luma scale, 2 chroma scale (GfwYuvPlane.fix).
__device__ __forceinline__ float fix_range1(float v, int fix, int c) {
    if (fix) { v = v * (fix == 1 ? 0.85882352f : 0.87843137f); if (c < 2) v = v + 16.0f; }
    return v;
}
template <typename T, int N>
__device__ __forceinline__ void store_px(uint8_t *dst, int off, const float *v, bool sat = true, int fix = 0) {
    T *d = reinterpret_cast<T *>(dst + (uint32_t)off);          // off >= 0: a zero-extended lane offset on the uniform plane base
    #pragma unroll
    for (int c = 0; c < N; ++c) {
        const float x = fix_range1(v[c], fix, c);
        T t;
        if (is_f32<T>::value) t = (T)x;                                         // f32 pixels pass through (pixel_formats.rs:247,296)
        else t = sat ? (T)gfw_f2u_sat(x, sizeof(T) == 1 ? 255.0f : 65535.0f) : (T)gfw_f2u_trunc(x);       // `as u8/u16`
        d[c] = t;
        gfw_ck<T>((uint32_t)off + (uint32_t)(c * sizeof(T)), t);
    }
}
// One plane.  32-bit byte offsets from the uniform plane base (planes are < 2 GiB, checked on the host).
template <typename T, int N, int I>
__device__ __forceinline__ void sample_store_bins(int bx, int by, bool ok, const GfwYuvPlane &P, const float *bg, float limit, int ox, int oy, const float *lut) {
    float out[N];
    #pragma unroll
    for (int c = 0; c < N; ++c) out[c] = bg[c];
    if (ok) {
        const Bins<I> b = bins_of<I>(bx, by, lut);
        if (__builtin_expect((bins_inside<T, N, I>(b, P.w, P.h)), 1))
            taps_inside<T, N, I>(P.src, row_off(b.sy, P.src_stride) + b.sx * (int)(N * sizeof(T)), P.src_stride, b, limit, out);
        else
            taps_edge<T, N, I>(P.src, P.src_stride, b, P.w, P.h, bg, limit, out);
    }
    store_px<T, N>(P.dst, row_off(oy, P.dst_stride) + ox * (int)(N * sizeof(T)), out, px_needs_sat<T>(bg, N, limit), P.fix);
}
template <typename T, int N, int I>
__device__ __forceinline__ void sample_store(float u, float v, bool ok, const GfwYuvPlane &P, const float *bg, float limit, int ox, int oy, const float *lut) {
    sample_store_bins<T, N, I>(bin_x<I>(u, P), bin_y<I>(v, P), ok, P, bg, limit, ox, oy, lut);       // (garbage bins of a point that is not ok are never used)
}
// Planar planes of identical geometry (U and V; or G,B,R,A of a planar float frame): one set of bins / weights /
// offsets, one gather set per plane.
template <typename T, int I>
__device__ __forceinline__ void sample_store_shared(float u, float v, bool ok, const GfwYuvPlane *pl, int first, int last, int ox, int oy, const float *lut) {
    const GfwYuvPlane &P0 = pl[first];
    Bins<I> b;
    b.sx = 0; b.sy = 0; b.tx = lut; b.ty = lut;
    bool inside = false;
    int off0 = 0;
    if (ok) {
        b = bins_of<I>(bin_x<I>(u, P0), bin_y<I>(v, P0), lut);
        inside = bins_inside<T, 1, I>(b, P0.w, P0.h);
        off0 = row_off(b.sy, P0.src_stride) + b.sx * (int)sizeof(T);
    }
    const int doff = row_off(oy, P0.dst_stride) + ox * (int)sizeof(T);
    #pragma unroll 1
    for (int pi = first; pi <= last; ++pi) {
        float o = pl[pi].bg[0];
        if (ok) {
            if (__builtin_expect(inside, 1)) taps_inside<T, 1, I>(pl[pi].src, off0, P0.src_stride, b, pl[pi].limit, &o);
            else taps_edge<T, 1, I>(pl[pi].src, P0.src_stride, b, P0.w, P0.h, pl[pi].bg, pl[pi].limit, &o);
        }
        store_px<T, 1>(pl[pi].dst, doff, &o, true, pl[pi].fix);
    }
}
// The same over named planes (baked builds: the planes are separate objects, never an array — an array indexed by a loop counter would
// live in scratch and turn the plane pointers into flat addresses).  n = 1..3 planes Pa, Pb, Pc.
template <typename T, int I>
__device__ __forceinline__ void sample_store_shared_refs(float u, float v, bool ok, const GfwYuvPlane &Pa, const GfwYuvPlane &Pb, const GfwYuvPlane &Pc, int n,
                                                         int ox, int oy, const float *lut) {
    Bins<I> b;
    b.sx = 0; b.sy = 0; b.tx = lut; b.ty = lut;
    bool inside = false;
    int off0 = 0;
    if (ok) {
        b = bins_of<I>(bin_x<I>(u, Pa), bin_y<I>(v, Pa), lut);
        inside = bins_inside<T, 1, I>(b, Pa.w, Pa.h);
        off0 = row_off(b.sy, Pa.src_stride) + b.sx * (int)sizeof(T);
    }
    const int doff = row_off(oy, Pa.dst_stride) + ox * (int)sizeof(T);
    auto one = [&](const GfwYuvPlane &P) {
        float o = P.bg[0];
        if (ok) {
            if (__builtin_expect(inside, 1)) taps_inside<T, 1, I>(P.src, off0, Pa.src_stride, b, P.limit, &o);
            else taps_edge<T, 1, I>(P.src, Pa.src_stride, b, Pa.w, Pa.h, P.bg, P.limit, &o);
        }
        store_px<T, 1>(P.dst, doff, &o, px_needs_sat<T>(P.bg, 1, P.limit), P.fix);
    };
    one(Pa);
    if (n > 1) one(Pb);
    if (n > 2) one(Pc);
}

// ---- integer-dot taps for 8/16-bit planes ---------------------------------------------------------------------------------
// A bilinear sample of an integer plane is sum = RN(RN(xs0*cy0) + RN(xs1*cy1)) with xs = p0*(1-k/32) + p1*k/32 exact
// (cpu_undistort.rs:392-411).  xs*32 = p0*(32-k) + p1*k is ONE integer dot instruction on the raw loaded word
// (v_dot2_u32_u16 / v_dot4_u32_u8), converted exactly (< 2^22), multiplied by the integer y weight and scaled by 2^-10 at the
// end — power-of-two scaling commutes with round-to-nearest, so the two roundings are the reference's.

template <typename T, bool UV> struct HotTap;
template <> struct HotTap<uint8_t, false> {                  // two u8 taps = one 16-bit word at any address
    static constexpr int BYTES = 2, PX = 1;
    typedef uint16_t u16u __attribute__((aligned(1)));
    static __device__ __forceinline__ uint32_t load(const uint8_t *src, uint32_t off) { return *reinterpret_cast<const u16u *>(src + off); }
    static __device__ __forceinline__ uint32_t wpack(uint32_t k) { return (32u - k) | (k << 8); }
    static __device__ __forceinline__ uint32_t dot(uint32_t raw, uint32_t w) { return __builtin_amdgcn_udot4(raw, w, 0u, false); }
};
// interleaved chroma: (U0 V0 U1 V1)
template <> struct HotTap<uint8_t, true> {                   // four bytes at a 2-byte aligned address
    static constexpr int BYTES = 4, PX = 2;
    typedef uint32_t u32u __attribute__((aligned(2)));
    static __device__ __forceinline__ uint32_t load(const uint8_t *src, uint32_t off) { return *reinterpret_cast<const u32u *>(src + off); }
    static __device__ __forceinline__ uint32_t wpack(uint32_t k) { return (32u - k) | (k << 16); }          // bytes 0 and 2 (U); << 8 for V
    static __device__ __forceinline__ void dot(uint32_t raw, uint32_t w, uint32_t &u, uint32_t &v) {
        u = __builtin_amdgcn_udot4(raw, w, 0u, false); v = __builtin_amdgcn_udot4(raw, w << 8, 0u, false);
    }
};
__device__ __forceinline__ uint32_t hot_f2u(float v) { uint32_t r; asm("v_cvt_u32_f32 %0, %1" : "=v"(r) : "v"(v)); return r; }
// RN(RN(xs0*cy0) + RN(xs1*cy1)) from the two integer row sums (see the header comment), clamped by pixel_value_limit
__device__ __forceinline__ uint32_t hot_blend(uint32_t i0, uint32_t i1, uint32_t ky, float limit) {
    const float s = ((float)i0 * (float)(32u - ky) + (float)i1 * (float)ky) * 0.0009765625f;
    return hot_f2u(min_limit(s, limit));
}

// ---- bilinear specialisation (I = 2): named weights, two-compare interior test — the hot configuration ---------------------------------------------------
struct Bins2 { int sx, sy; float cx0, cx1, cy0, cy1; uint32_t kx, ky; };
__device__ __forceinline__ Bins2 bins2_of(int sx0, int sy0) {             // from the two rounded 1/32-pixel coordinates
    Bins2 b;
    b.sx = sx0 >> 5; b.sy = sy0 >> 5;
    b.kx = (uint32_t)sx0 & 31u; b.ky = (uint32_t)sy0 & 31u;
    b.cx1 = (float)(sx0 & 31) * 0.03125f; b.cx0 = 1.0f - b.cx1;     // {1-k/32, k/32}: the LUT row (cpu_undistort.rs:14-19)
    b.cy1 = (float)(sy0 & 31) * 0.03125f; b.cy0 = 1.0f - b.cy1;
    return b;
}
__device__ __forceinline__ Bins2 make_bins2(float u, float v) { return bins2_of(round32_i32(u), round32_i32(v)); }
// Taps that straddle the source rect: out-of-rect taps read `bg`, out-of-rect rows contribute bg*cy
// (cpu_undistort.rs:392-409), in the reference's exact operation order.
template <typename T, int N>
__device__ __forceinline__ void taps_edge2(const uint8_t *src, int stride, const Bins2 &b, int w, int h, const float *bg, float limit, float *out) {
    const bool x0in = b.sx >= 0 && b.sx < w, x1in = b.sx + 1 >= 0 && b.sx + 1 < w;
    const bool y0in = b.sy >= 0 && b.sy < h, y1in = b.sy + 1 >= 0 && b.sy + 1 < h;
    const T *row0 = reinterpret_cast<const T *>(src + (int64_t)b.sy * stride) + (int64_t)b.sx * N;
    const T *row1 = reinterpret_cast<const T *>(reinterpret_cast<const uint8_t *>(row0) + stride);
    #pragma unroll
    for (int c = 0; c < N; ++c) {
        const float p00 = (y0in && x0in) ? (float)row0[c] : bg[c];
        const float p01 = (y0in && x1in) ? (float)row0[N + c] : bg[c];
        const float p10 = (y1in && x0in) ? (float)row1[c] : bg[c];
        const float p11 = (y1in && x1in) ? (float)row1[N + c] : bg[c];
        float sum = 0.0f;
        if (y0in) { float xs = 0.0f; xs = xs + p00 * b.cx0; xs = xs + p01 * b.cx1; sum = sum + xs * b.cy0; } else sum = sum + bg[c] * b.cy0;
        if (y1in) { float xs = 0.0f; xs = xs + p10 * b.cx0; xs = xs + p11 * b.cx1; sum = sum + xs * b.cy1; } else sum = sum + bg[c] * b.cy1;
        out[c] = fminf(sum, limit);
    }
}
// All four taps inside.  For the integer pixel types every tap is >= +0, so the reference's leading zero-adds
// (xsum = 0 + p*c, sum = 0 + xs*cy) are exact identities and are dropped; for f32 pixels (-0, negative values) they stay.
template <typename T, int N, bool CAST_FOLLOWS = false>     // CAST_FOLLOWS: the caller truncates the value to the integer type next (no colour-range fix in between)
__device__ __forceinline__ void taps_inside2(const uint8_t *src, int off0, int stride, const Bins2 &b, float limit, float *out) {
    // all four taps inside: off0 >= 0 and the plane is < 2 GiB (host-checked), so both rows are zero-extended 32-bit lane offsets on the
    // uniform plane base — the saddr + voffset form of global_load, one address instruction per row instead of a 64-bit add chain
    // (the two taps of a 16-bit row as the ALIGNED dwordx2 around them + v_alignbit — the form the bicubic / Lanczos4 rows need — measured 42.3 = 42.4 us per C2 frame:
    //  neighbouring lanes' fetches coalesce either way; profiles/r05_c2_memory_path.txt)
    uint32_t off1 = (uint32_t)off0 + (uint32_t)stride;
    asm("" : "+v"(off1));             // opaque: keeps the second row a 32-bit lane offset too (a pitch beyond the 12-bit immediate otherwise becomes a 64-bit add chain)
    const T *row0 = reinterpret_cast<const T *>(src + (uint32_t)off0);
    const T *row1 = reinterpret_cast<const T *>(src + off1);
    #pragma unroll
    for (int c = 0; c < N; ++c) {
        if (is_f32<T>::value) {
            float xs0 = 0.0f; xs0 = xs0 + (float)row0[c] * b.cx0; xs0 = xs0 + (float)row0[N + c] * b.cx1;
            float xs1 = 0.0f; xs1 = xs1 + (float)row1[c] * b.cx0; xs1 = xs1 + (float)row1[N + c] * b.cx1;
            float sum = 0.0f; sum = sum + xs0 * b.cy0; sum = sum + xs1 * b.cy1;
            out[c] = fminf(sum, limit);
        } else {
            // tap (<= 16 bits) x weight (k/32) and the sum of two such products are exact in f32 (<= 22 bits), so the
            // fused form rounds nowhere the reference's separate multiply and add would
            const float xs0 = __builtin_fmaf((float)row0[N + c], b.cx1, (float)row0[c] * b.cx0);
            const float xs1 = __builtin_fmaf((float)row1[N + c], b.cx1, (float)row1[c] * b.cx0);
            // min(sum, pixel_value_limit) ahead of a truncating cast is idle when the limit is the type's maximum: the weights of a row and of the two rows sum to 1 exactly
            // (k/32 and 1 - k/32), so the real sum is <= 65535 (255), each product and the sum round up by at most one part in 2^24: sum <= 65535.004 < 65536, and the cast
            // truncates.  Known at compile time in a baked build only.
            const float sum = xs0 * b.cy0 + xs1 * b.cy1;
            out[c] = (GFW_BAKE && CAST_FOLLOWS && limit >= (sizeof(T) == 1 ? 255.0f : 65535.0f)) ? sum : min_limit(sum, limit);
        }
    }
}
// One plane.  32-bit byte offsets from the uniform plane base (planes are < 2 GiB, checked on the host).
// Audit mode (aud != nullptr, a compile-time constant after inlining): every byte range about to be touched is checked
// against the length the caller declared for the buffer; violations are counted in aud[5] and the access is skipped.
__device__ __forceinline__ bool range_ok(unsigned long long *aud, int64_t off, int64_t bytes, int len) {
    if (!aud) return true;
    if (off >= 0 && off + bytes <= (int64_t)len) return true;
    atomicAdd(&aud[5], 1ull);
    return false;
}
template <typename T, int N>
__device__ __forceinline__ void sample_store2_bins(int bx, int by, bool ok, const GfwYuvPlane &P, const float *bg, float limit, int ox, int oy,
                                                   unsigned long long *aud = nullptr) {
    float out[N];
    #pragma unroll
    for (int c = 0; c < N; ++c) out[c] = bg[c];
    if (ok) {
        const Bins2 b = bins2_of(bx, by);
        if (__builtin_expect((unsigned)b.sx < (unsigned)(P.w - 1) && (unsigned)b.sy < (unsigned)(P.h - 1), 1)) {
            const int off0 = row_off(b.sy, P.src_stride) + b.sx * (int)(N * sizeof(T));
            if (range_ok(aud, off0, 2 * N * sizeof(T), P.src_len) && range_ok(aud, (int64_t)off0 + P.src_stride, 2 * N * sizeof(T), P.src_len)) {
                if constexpr (!is_f32<T>::value && (N == 1 || N == 2) && sizeof(T) == 1) { if (!P.fix) {      // (the range fix sits between the blend and the cast: the float path below)
                    // integer-dot taps: the pixel value comes out as an integer; store it and leave
                    const uint32_t doff = (uint32_t)oy * (uint32_t)P.dst_stride + (uint32_t)ox * (uint32_t)(N * sizeof(T));
                    if (!range_ok(aud, doff, N * sizeof(T), P.dst_len)) return;
                    if constexpr (N == 1) {
                        typedef HotTap<T, false> Tap;
                        const uint32_t r0 = Tap::load(P.src, (uint32_t)off0), r1 = Tap::load(P.src, (uint32_t)off0 + (uint32_t)P.src_stride);
                        const uint32_t w = Tap::wpack(b.kx);
                        const T o1 = (T)hot_blend(Tap::dot(r0, w), Tap::dot(r1, w), b.ky, limit);
                        *reinterpret_cast<T *>(P.dst + doff) = o1;
                        gfw_ck<T>(doff, o1);
                    } else {
                        typedef HotTap<T, true> Tap;
                        const auto r0 = Tap::load(P.src, (uint32_t)off0), r1 = Tap::load(P.src, (uint32_t)off0 + (uint32_t)P.src_stride);
                        const uint32_t w = Tap::wpack(b.kx);
                        uint32_t u0, v0, u1, v1;
                        Tap::dot(r0, w, u0, v0); Tap::dot(r1, w, u1, v1);
                        const uint32_t ou = hot_blend(u0, u1, b.ky, limit), ov = hot_blend(v0, v1, b.ky, limit);
                        T *d = reinterpret_cast<T *>(P.dst + doff);
                        d[0] = (T)ou; d[1] = (T)ov;
                        gfw_ck_pair<T>(doff, ou, ov);
                    }
                    return;
                } }
                taps_inside2<T, N>(P.src, off0, P.src_stride, b, limit, out);
            }
        } else
            taps_edge2<T, N>(P.src, P.src_stride, b, P.w, P.h, bg, limit, out);
    }
    const int doff = row_off(oy, P.dst_stride) + ox * (int)(N * sizeof(T));
    if (range_ok(aud, doff, N * sizeof(T), P.dst_len)) store_px<T, N>(P.dst, doff, out, px_needs_sat<T>(bg, N, limit), P.fix);
}
template <typename T, int N>
__device__ __forceinline__ void sample_store2(float u, float v, bool ok, const GfwYuvPlane &P, const float *bg, float limit, int ox, int oy,
                                              unsigned long long *aud = nullptr) {
    sample_store2_bins<T, N>(bin_x<2>(u, P), bin_y<2>(v, P), ok, P, bg, limit, ox, oy, aud);       // (garbage bins of a point that is not ok are never used)
}
// Planar planes of identical geometry (U and V; or G,B,R,A of a planar float frame): one set of bins / weights /
// offsets, one gather pair per plane.
template <typename T>
__device__ __forceinline__ void sample_store_shared2(float u, float v, bool ok, const GfwYuvPlane *pl, int first, int last, int ox, int oy) {
    const GfwYuvPlane &P0 = pl[first];
    Bins2 b = {0, 0, 0.0f, 0.0f, 0.0f, 0.0f, 0u, 0u};
    bool inside = false;
    int off0 = 0;
    if (ok) {
        b = bins2_of(bin_x<2>(u, P0), bin_y<2>(v, P0));
        inside = (unsigned)b.sx < (unsigned)(P0.w - 1) && (unsigned)b.sy < (unsigned)(P0.h - 1);
        off0 = row_off(b.sy, P0.src_stride) + b.sx * (int)sizeof(T);
    }
    const int doff = row_off(oy, P0.dst_stride) + ox * (int)sizeof(T);
    #pragma unroll 1
    for (int pi = first; pi <= last; ++pi) {
        float o = pl[pi].bg[0];
        if (ok) {
            if (__builtin_expect(inside, 1)) taps_inside2<T, 1>(pl[pi].src, off0, P0.src_stride, b, pl[pi].limit, &o);
            else taps_edge2<T, 1>(pl[pi].src, P0.src_stride, b, P0.w, P0.h, pl[pi].bg, pl[pi].limit, &o);
        }
        store_px<T, 1>(pl[pi].dst, doff, &o, true, pl[pi].fix);
    }
}
template <typename T>
__device__ __forceinline__ void sample_store_shared2_refs(float u, float v, bool ok, const GfwYuvPlane &Pa, const GfwYuvPlane &Pb, const GfwYuvPlane &Pc, int n, int ox, int oy) {
    Bins2 b = {0, 0, 0.0f, 0.0f, 0.0f, 0.0f, 0u, 0u};
    bool inside = false;
    int off0 = 0;
    if (ok) {
        b = bins2_of(bin_x<2>(u, Pa), bin_y<2>(v, Pa));
        inside = (unsigned)b.sx < (unsigned)(Pa.w - 1) && (unsigned)b.sy < (unsigned)(Pa.h - 1);
        off0 = row_off(b.sy, Pa.src_stride) + b.sx * (int)sizeof(T);
    }
    const int doff = row_off(oy, Pa.dst_stride) + ox * (int)sizeof(T);
    // every plane's taps first, then the stores: a store between two planes' fetches (the destination may alias a source as far as the compiler knows) had them
    // issued plane by plane — fetch, wait, blend, store, four times over for planar float frames (the disassembly of the shipped C4 kernel, round 4)
    float oa = Pa.bg[0], ob = n > 1 ? Pb.bg[0] : 0.0f, oc = n > 2 ? Pc.bg[0] : 0.0f;
    if (ok) {
        if (__builtin_expect(inside, 1)) {
            taps_inside2<T, 1>(Pa.src, off0, Pa.src_stride, b, Pa.limit, &oa);
            if (n > 1) taps_inside2<T, 1>(Pb.src, off0, Pa.src_stride, b, Pb.limit, &ob);
            if (n > 2) taps_inside2<T, 1>(Pc.src, off0, Pa.src_stride, b, Pc.limit, &oc);
        } else {
            taps_edge2<T, 1>(Pa.src, Pa.src_stride, b, Pa.w, Pa.h, Pa.bg, Pa.limit, &oa);
            if (n > 1) taps_edge2<T, 1>(Pb.src, Pa.src_stride, b, Pa.w, Pa.h, Pb.bg, Pb.limit, &ob);
            if (n > 2) taps_edge2<T, 1>(Pc.src, Pa.src_stride, b, Pa.w, Pa.h, Pc.bg, Pc.limit, &oc);
        }
    }
    store_px<T, 1>(Pa.dst, doff, &oa, px_needs_sat<T>(Pa.bg, 1, Pa.limit), Pa.fix);
    if (n > 1) store_px<T, 1>(Pb.dst, doff, &ob, px_needs_sat<T>(Pb.bg, 1, Pb.limit), Pb.fix);
    if (n > 2) store_px<T, 1>(Pc.dst, doff, &oc, px_needs_sat<T>(Pc.bg, 1, Pc.limit), Pc.fix);
}

// Two planar chroma planes of identical geometry (U, V) — the C2 hot path: one set of bins / weights / offsets,
// two gathers, no loop over a plane index (which would index the kernel-argument plane array dynamically).
template <typename T>
__device__ __forceinline__ void sample_store_uv2(float u, float v, bool ok, const GfwYuvPlane &PU, const GfwYuvPlane &PV,
                                                 float bg_u, float bg_v, float lim_u, float lim_v, int ox, int oy, unsigned long long *aud = nullptr) {
    float ou = bg_u, ov = bg_v;
    if (ok) {
        const Bins2 b = bins2_of(bin_x<2>(u, PU), bin_y<2>(v, PU));
        if (__builtin_expect((unsigned)b.sx < (unsigned)(PU.w - 1) && (unsigned)b.sy < (unsigned)(PU.h - 1), 1)) {
            const int off0 = row_off(b.sy, PU.src_stride) + b.sx * (int)sizeof(T);
            const int top = PU.src_len < PV.src_len ? PU.src_len : PV.src_len;
            if (range_ok(aud, off0, 2 * sizeof(T), top) && range_ok(aud, (int64_t)off0 + PU.src_stride, 2 * sizeof(T), top)) {
                if constexpr (!is_f32<T>::value && sizeof(T) == 1) { if (!PU.fix && !PV.fix) {
                    typedef HotTap<T, false> Tap;
                    const uint32_t doff = (uint32_t)oy * (uint32_t)PU.dst_stride + (uint32_t)ox * (uint32_t)sizeof(T);
                    if (!range_ok(aud, doff, sizeof(T), PU.dst_len < PV.dst_len ? PU.dst_len : PV.dst_len)) return;
                    const uint32_t a0 = Tap::load(PU.src, (uint32_t)off0), a1 = Tap::load(PU.src, (uint32_t)off0 + (uint32_t)PU.src_stride);
                    const uint32_t b0 = Tap::load(PV.src, (uint32_t)off0), b1 = Tap::load(PV.src, (uint32_t)off0 + (uint32_t)PU.src_stride);
                    const uint32_t w = Tap::wpack(b.kx);
                    const T o_u = (T)hot_blend(Tap::dot(a0, w), Tap::dot(a1, w), b.ky, lim_u), o_v = (T)hot_blend(Tap::dot(b0, w), Tap::dot(b1, w), b.ky, lim_v);
                    *reinterpret_cast<T *>(PU.dst + doff) = o_u;
                    *reinterpret_cast<T *>(PV.dst + doff) = o_v;
                    gfw_ck<T>(doff, o_u); gfw_ck<T>(doff, o_v);
                    return;
                } }
                taps_inside2<T, 1>(PU.src, off0, PU.src_stride, b, lim_u, &ou);
                taps_inside2<T, 1>(PV.src, off0, PU.src_stride, b, lim_v, &ov);
            }
        } else {
            taps_edge2<T, 1>(PU.src, PU.src_stride, b, PU.w, PU.h, &bg_u, lim_u, &ou);
            taps_edge2<T, 1>(PV.src, PU.src_stride, b, PU.w, PU.h, &bg_v, lim_v, &ov);
        }
    }
    const int doff = row_off(oy, PU.dst_stride) + ox * (int)sizeof(T);
    if (!range_ok(aud, doff, sizeof(T), PU.dst_len < PV.dst_len ? PU.dst_len : PV.dst_len)) return;
    store_px<T, 1>(PU.dst, doff, &ou, px_needs_sat<T>(&bg_u, 1, lim_u), PU.fix);
    store_px<T, 1>(PV.dst, doff, &ov, px_needs_sat<T>(&bg_v, 1, lim_v), PV.fix);
}

// ---- branch-free sampling of a lane-row whose every tap is inside (round 4) -------------------------------------------------
// The phase-3 loop asks the wave once whether all its samples of a row are interior (`__all`, a uniform branch) and then runs these: no `ok` /
// inside / edge tests and no exec bookkeeping — the same loads, weights and operations as the interior branches of sample_store2 /
// sample_store_uv2.  Waves that touch the frame border, background or an invalid point take those functions as before.
__device__ __forceinline__ bool bins2_inside(const Bins2 &b, int w, int h) { return (unsigned)b.sx < (unsigned)(w - 1) && (unsigned)b.sy < (unsigned)(h - 1); }
// the value of one interior sample of a single-channel plane, converted like `as u8 / u16` (integer types) or as its f32 bit pattern
// (16-bit taps through v_dot2_u32_u16, the form the 8-bit planes use with v_dot4: 15 fewer vector instructions per pixel pair and 46.8 instead of 45.8 us per C2
// frame — measured in round 4 as in round 2, not kept: profiles/r04_ab_fastrow.txt)
template <typename T>
__device__ __forceinline__ uint32_t inside_value1(const uint8_t *src, int stride, const Bins2 &b, const float *bg, float limit) {
    const int off0 = row_off(b.sy, stride) + b.sx * (int)sizeof(T);
    if constexpr (!is_f32<T>::value && sizeof(T) == 1) {
        typedef HotTap<T, false> Tap;
        const uint32_t r0 = Tap::load(src, (uint32_t)off0), r1 = Tap::load(src, (uint32_t)off0 + (uint32_t)stride);
        const uint32_t w = Tap::wpack(b.kx);
        return hot_blend(Tap::dot(r0, w), Tap::dot(r1, w), b.ky, limit);
    } else {
        float o;
        taps_inside2<T, 1, true>(src, off0, stride, b, limit, &o);
        if constexpr (is_f32<T>::value) return gfw_f2u(o);
        else return px_needs_sat<T>(bg, 1, limit) ? gfw_f2u_sat(o, 65535.0f) : gfw_f2u_trunc(o);
    }
}
// ... and of a bicubic / Lanczos4 sample (the interior taps of taps_inside)
template <typename T, int I>
__device__ __forceinline__ uint32_t inside_value1_lut(const uint8_t *src, int stride, int bx, int by, const float *bg, float limit, const float *lut) {
    const Bins<I> b = bins_of<I>(bx, by, lut);
    float o;
    taps_inside<T, 1, I>(src, row_off(b.sy, stride) + b.sx * (int)sizeof(T), stride, b, limit, &o);
    if constexpr (is_f32<T>::value) return gfw_f2u(o);
    else return px_needs_sat<T>(bg, 1, limit) ? gfw_f2u_sat(o, sizeof(T) == 1 ? 255.0f : 65535.0f) : gfw_f2u_trunc(o);
}
template <typename T, int I>
__device__ __forceinline__ GfwVote lut_interior(int bx, int by, int w, int h) {          // bins_inside, as a vote over its terms
    constexpr int TAP_MARGIN = tap_margin<T, 1, I>();
    if (!(w >= I + TAP_MARGIN && h >= I)) return gfw_lanes(false);
    return gfw_lanes((unsigned)(bx >> 5) <= (unsigned)(w - I - TAP_MARGIN)) & gfw_lanes((unsigned)(by >> 5) <= (unsigned)(h - I));
}
template <typename T>
__device__ __forceinline__ void store_value1(uint8_t *dst, int off, uint32_t v, unsigned long long *ck = nullptr) {
    if constexpr (is_f32<T>::value && sizeof(T) == 4) { *reinterpret_cast<uint32_t *>(dst + (uint32_t)off) = v; gfw_ck<uint32_t>((uint32_t)off, v, ck); }
    else if constexpr (is_f32<T>::value) { const T h = (T)__builtin_bit_cast(float, v); *reinterpret_cast<T *>(dst + (uint32_t)off) = h; gfw_ck<T>((uint32_t)off, h, ck); }      // (f16 planes: the f32 bit pattern, narrowed)
    else { *reinterpret_cast<T *>(dst + (uint32_t)off) = (T)v; gfw_ck<T>((uint32_t)off, (T)v, ck); }
}
// two horizontally adjacent samples of an integer plane leave as ONE store (the pair's address need not be aligned to the pair: global memory takes it)
template <typename T>
__device__ __forceinline__ void store_pair1(uint8_t *dst, int off, uint32_t v0, uint32_t v1, unsigned long long *ck = nullptr, bool ck_inside = false) {
    if constexpr (sizeof(T) == 1) { typedef uint16_t u16u __attribute__((aligned(1))); *reinterpret_cast<u16u *>(dst + (uint32_t)off) = (uint16_t)(v0 | (v1 << 8)); }
    else if constexpr (sizeof(T) == 2) { typedef uint32_t u32u __attribute__((aligned(2))); *reinterpret_cast<u32u *>(dst + (uint32_t)off) = v0 | (v1 << 16); }
    else { typedef uint2 u2u __attribute__((aligned(4))); *reinterpret_cast<u2u *>(dst + (uint32_t)off) = uint2{v0, v1}; }
    if constexpr (sizeof(T) == 1) gfw_ck_pair<uint8_t>((uint32_t)off, v0, v1, ck, ck_inside);
    else if constexpr (sizeof(T) == 2) gfw_ck_pair<uint16_t>((uint32_t)off, v0, v1, ck, ck_inside);
    else gfw_ck_pair<uint32_t>((uint32_t)off, v0, v1, ck, ck_inside);
}

// ---- background mode 3 ("margin with feather", cpu_undistort.rs:576-613) ----------------------------------------------------
// Near the frame border the pixel is c1 * alpha + c2 * (1 - alpha): c1 sampled at the projected point, c2 at the point pulled
// towards the centre by background_margin, alpha the distance to the border in units of the feather.  uv lives in full-resolution
// coordinates for every plane, so alpha and the second point are the same for a luma pixel and the chroma site that shares its
// coordinate.  Served by the GFW_MODEL_GENERIC_EXTRA instantiation only (extras & 16).
struct Feather { float alpha, x2, y2; };
__device__ __forceinline__ Feather feather_of(float ux, float uy, const GfwYuvArgs &A) {
    const float width_f = (float)AF(width), height_f = (float)AF(height);
    const float widthf = width_f - 1.0f, heightf = height_f - 1.0f;
    const float feather = fmaxf(A.kp.background_margin_feather * heightf, 0.0001f);
    Feather f{1.0f, ux, uy};
    if ((ux > widthf - feather) || (ux < feather) || (uy > heightf - feather) || (uy < feather)) {
        f.alpha = fmaxf(fminf(fminf(fminf(fminf(widthf - ux, heightf - uy), ux), uy) / feather, 1.0f), 0.0f);
        float p2x = ux / width_f, p2y = uy / height_f;
        p2x = ((p2x - 0.5f) * (1.0f - A.kp.background_margin)) + 0.5f;
        p2y = ((p2y - 0.5f) * (1.0f - A.kp.background_margin)) + 0.5f;
        f.x2 = p2x * width_f; f.y2 = p2y * height_f;
    }
    return f;
}
// sample_input_at for the LUT samplers (cpu_undistort.rs:371-418) without the store: N channels of one plane at (u, v).
template <typename T, int N, int I>
__device__ __forceinline__ void sample_only(float u, float v, const GfwYuvPlane &P, const float *bg, float limit, const float *lut, float *out) {
    if (I == 2) {
        const Bins2 b = bins2_of(bin_x<2>(u, P), bin_y<2>(v, P));
        if ((unsigned)b.sx < (unsigned)(P.w - 1) && (unsigned)b.sy < (unsigned)(P.h - 1))
            taps_inside2<T, N>(P.src, row_off(b.sy, P.src_stride) + b.sx * (int)(N * sizeof(T)), P.src_stride, b, limit, out);
        else
            taps_edge2<T, N>(P.src, P.src_stride, b, P.w, P.h, bg, limit, out);
    } else {
        const Bins<I> b = bins_of<I>(bin_x<I>(u, P), bin_y<I>(v, P), lut);
        if (bins_inside<T, N, I>(b, P.w, P.h))
            taps_inside<T, N, I>(P.src, row_off(b.sy, P.src_stride) + b.sx * (int)(N * sizeof(T)), P.src_stride, b, limit, out);
        else
            taps_edge<T, N, I>(P.src, P.src_stride, b, P.w, P.h, bg, limit, out);
    }
}
// One plane's pixel in background mode 3: two samples, blended, stored.  (mul_x, mul_y) = the plane's source_rect map.
template <typename T, int N, int I, bool INF_SAFE>
__device__ __forceinline__ void feather_store(float ux, float uy, const Feather &f, const GfwYuvPlane &P, const float *bg, float limit,
                                              float mul_x, float mul_y, const Maps &MP, int ox, int oy, const float *lut) {
    float c1[N], c2[N], px[N];
    sample_only<T, N, I>(map_c<INF_SAFE>(ux, mul_x, MP.den_x, MP.rcp_x), map_c<INF_SAFE>(uy, mul_y, MP.den_y, MP.rcp_y), P, bg, limit, lut, c1);
    sample_only<T, N, I>(map_c<INF_SAFE>(f.x2, mul_x, MP.den_x, MP.rcp_x), map_c<INF_SAFE>(f.y2, mul_y, MP.den_y, MP.rcp_y), P, bg, limit, lut, c2);
    #pragma unroll
    for (int c = 0; c < N; ++c) px[c] = c1[c] * f.alpha + c2[c] * (1.0f - f.alpha);
    store_px<T, N>(P.dst, row_off(oy, P.dst_stride) + ox * (int)(N * sizeof(T)), px, true, P.fix);
}



// ---- first pass (rolling-shutter row pick) -----------------------------------------------------------------
// The mid-row projection of undistort_coord (cpu_undistort.rs:470-479) is used for ONE thing: the integer
// sy = clamp(round(p.y)).  FAST1 evaluates p.y with fused arithmetic and a per-lens table of
// s(rho) = theta_d(atan(sqrt(rho)))/sqrt(rho) (linear interpolation, rho = (X/W)^2 + (Y/W)^2) and accepts the
// rounded value only when no half-integer lies within +-E of it, E bounding |approx - exact| (derivation in
// DESIGN.md section 2; tests/test_gpu_pass1.py audits every certificate and measures the real gap).  Everything
// else — a percent or two of the pixels — goes through the exact projection: queued in LDS per wave and resolved
// densely (one exact pass per few rows of the wave instead of one per pixel row).
struct Mid { float m0, m1, m2, m3, m4, m5, m6, m7, m8; };
struct P1 { float rho_max, rho_scale, eps, f, c, lim, wmin, rho_lim, gap, kmax; };     // rho_lim, gap (= 1/2 - E): the lattice form's; kmax: the table's last key (rho_max, or r_max in the r form)

template <int MODEL>
__device__ __forceinline__ int default_row(float ox, float oy, const GfwYuvArgs &A) {
    const int lim = AF(hrs) ? AF(width) : AF(height);
    return max(min(round_i32(AF(hrs) ? ox : oy), lim), 0);
}
// exact: cpu_undistort.rs:465-479
template <int MODEL>
__device__ __forceinline__ int pass1_exact(float ox, float oy, const Mid &M, const float *matrices, const Lens &L, const GfwYuvArgs &A) {
    int sy = default_row<MODEL>(ox, oy, A);
    const GfwPt pt = rd<MODEL>(ox, oy, float4{M.m0, M.m1, M.m2, M.m3}, float4{M.m4, M.m5, M.m6, M.m7}, M.m8, matrices + (size_t)(AF(matrix_count) / 2) * GFW_MAT_STRIDE + 8, L, A);
    if (pt.ok) { const int lim = AF(hrs) ? AF(width) : AF(height); sy = max(min(round_i32(AF(hrs) ? pt.x : pt.y), lim), 0); }
    return sy;
}
// The table's key.  The fisheye's scale theta_d(atan r) / r is a smooth function of rho = r^2 (theta_d is odd in theta) and is tabulated over rho: no square root.
// The radial models served since round 6 (GoPro's inverted polynomial, gopro.rs:25-72: even powers of the radius parameter) are smooth in r but have a sqrt(rho)
// kink at the optical centre as functions of rho: their tables run over r, and the first pass takes the hardware's square root (1 ulp: inside E).  GFW_P1_RFORM is
// a literal of a specialised build — the only builds that certify those models.
#ifndef GFW_P1_RFORM
#define GFW_P1_RFORM 0
#endif
// The key, clamped to the table's last one (a NaN becomes kmax: the hardware minimum).  In the r form the clamp is the COMPILER's v_min_f32, not min_limit's
// inline asm: gfx950 needs a wait state between a transcendental's result and a VALU instruction that reads it, the compiler's hazard recogniser inserts it for its
// own instructions and cannot see inside an asm statement — `v_sqrt_f32` followed at once by the asm's `v_min_f32` read a stale register (audit on the MI355X:
// gaps of 39 px, a third of the certificates wrong, varying with the optimisation level; the interpreter, which has no pipeline, was clean: profiles/r06_gopro_first_pass.txt)
__device__ __forceinline__ float p1_key_clamped(float rho, float kmax) {
    if (GFW_P1_RFORM) return __builtin_fminf(gfw_hw_sqrt(rho), kmax);
    return min_limit(rho, kmax);
}
// approximate + certificate; returns false when the exact path must decide.
// (ax, ay, aw) = ox*m0+m2, ox*m3+m5, ox*m6+m8 are per-lane constants of the pixel column.
__device__ __forceinline__ bool pass1_fast(float ax, float ay, float aw, float oy, const Mid &M, const P1 &Q, const float2 *tab,
                                           bool hrs, float rl2, int &sy, float &v_out, unsigned long long *aud = nullptr) {
    const float X = __builtin_fmaf(oy, M.m1, ax);
    const float Y = __builtin_fmaf(oy, M.m4, ay);
    const float W = __builtin_fmaf(oy, M.m7, aw);
    const float rw = gfw_hw_rcp(W);
    const float a = X * rw, b = Y * rw;
    const float rho = __builtin_fmaf(a, a, b * b);
    // W safely positive (the exact path decides validity otherwise) and rho inside the table (NaN fails both)
    bool good = (W > Q.wmin) & (rho < Q.rho_max);
    if (rl2 > 0.0f) {                                              // :139 — decide only when clear of the boundary
        const float lhs = __builtin_fmaf(X, X, Y * Y), rhs = rl2 * W;
        good = good & (lhs < rhs * 0.9999f);
    }
    // table position, clamped so that a rejected lane still indexes the table: rho is a sum of squares (>= +0, or NaN, which the
    // hardware minimum turns into rho_max); the interval index is the truncated position and v_fract_f32 its exact remainder
    const float tpos = p1_key_clamped(rho, Q.kmax) * Q.rho_scale;
    const uint32_t ti = gfw_f2u_trunc(tpos);
    if (aud && !(ti <= (uint32_t)GFW_P1_TABLE_N)) atomicAdd(&aud[5], 1ull);
    const float2 e = *reinterpret_cast<const float2 *>(reinterpret_cast<const uint8_t *>(tab) + ti * 8u);
    const float s = __builtin_fmaf(__builtin_amdgcn_fractf(tpos), e.y, e.x);
    const float v = __builtin_fmaf((hrs ? a : b) * s, Q.f, Q.c);
    v_out = v;
    const float g = v - 0.5f;
    const float dist = fabsf(g - rintf(g));                        // distance of v to the nearest half-integer
    const bool outside = (v <= -0.25f) | (v >= Q.lim + 0.25f);     // there the clamp decides and ties cannot matter (false for NaN)
    good = good & (outside | (dist > Q.eps));                      // a NaN fails both comparisons
    sy = max(min(gfw_f2i(rintf(v)), (int)Q.lim), 0);
    return good;
}

// The lattice form (round 5): the same approximate value at ONE point — a node of the wave's lattice — or NaN where the certificate's premises fail there
// (W not safely positive, rho beyond the table less the cell's margin: a pixel is certified only inside a cell whose four nodes are sound, and W, affine,
// and rho, Lipschitz by the margin, then satisfy the premises at every point of the cell).  Operation for operation pass1_fast's value.
__device__ __forceinline__ float pass1_node(float ox, float oy, const Mid &M, const P1 &Q, const float2 *tab, bool hrs) {
    const float X = __builtin_fmaf(oy, M.m1, __builtin_fmaf(ox, M.m0, M.m2));
    const float Y = __builtin_fmaf(oy, M.m4, __builtin_fmaf(ox, M.m3, M.m5));
    const float W = __builtin_fmaf(oy, M.m7, __builtin_fmaf(ox, M.m6, M.m8));
    const float rw = gfw_hw_rcp(W);
    const float a = X * rw, b = Y * rw;
    const float rho = __builtin_fmaf(a, a, b * b);
    const bool good = (W > Q.wmin) & (rho < Q.rho_lim);            // (a NaN fails both)
    const float tpos = p1_key_clamped(rho, Q.kmax) * Q.rho_scale;
    const uint32_t ti = gfw_f2u_trunc(tpos);
    const float2 e = *reinterpret_cast<const float2 *>(reinterpret_cast<const uint8_t *>(tab) + ti * 8u);
    const float s = __builtin_fmaf(__builtin_amdgcn_fractf(tpos), e.y, e.x);
    const float v = __builtin_fmaf((hrs ? a : b) * s, Q.f, Q.c);
    return good ? v : __builtin_nanf("");
}

#if GFW_TIMELINE
}  // namespace
extern "C" { __device__ unsigned long long gfw_tl[8192 * 8]; }      // external name: the host reads it by symbol (hipModuleGetGlobal in a run-time build)
#if GFW_TIMELINE >= 2
// GFW_TIMELINE = 2: where a wave's life goes INSIDE the branch-free row (round 5): per wave, accumulated shader clocks of [row start .. matrix rows arrived],
// [.. projection done], [.. luma taps issued and stored], [.. chroma done]; [4] rows counted.  A forced `s_waitcnt vmcnt(0)` separates the first two (diagnosis only).
extern "C" { __device__ unsigned long long gfw_tl_blocks[8192 * 8]; }
#define GFW_TLB(k) do { __builtin_amdgcn_sched_barrier(0); const unsigned long long tlb_now_ = __builtin_readcyclecounter(); tl_blk[k] += tlb_now_ - tl_mark; tl_mark = tlb_now_; __builtin_amdgcn_sched_barrier(0); } while (0)
#define GFW_TLB_START() do { __builtin_amdgcn_sched_barrier(0); tl_mark = __builtin_readcyclecounter(); __builtin_amdgcn_sched_barrier(0); } while (0)
#endif
namespace {
#endif
// The kernel body.  `clip` (baked builds only): the per-frame pointers of the frames of one launch — the frames of a clip share every
// other argument, so a launch can carry several of them and the occupancy tail of one frame is filled by the next (the effect two
// HIP streams showed: 79.3 -> 71.7 us per C2 frame, profiles/r03_ab_northstar.txt) without a second stream or a second launch.
template <int MODEL, typename T, int N0, int I, int DW, int DH, bool INTERLEAVED_UV, int RB, bool FAST1, bool AUDIT>
__device__ __forceinline__ void gfw_yuv_body(const GfwYuvArgs &A_in, const GfwClipArgs *clip) {
    // A baked build (run time, gfw_jit.hip) reads every clip-invariant argument as a literal from the bake header (AF(x) = GFW_BK_x): the
    // loads, the uniform branches and the scalar registers they pin disappear — the reference bakes its per-clip constants into the
    // OpenCL source it compiles per clip the same way (opencl.rs:181-214).  Pointers and the per-frame fields stay arguments.
    const GfwYuvArgs &A = A_in;
#if GFW_TIMELINE
    const unsigned long long tl_start = wall_clock64();           // the wave's first instruction
#endif
#if GFW_BAKE
    // (`clip` is the kernel's own by-value argument in a baked build, never null — and must not be TESTED: a pointer comparison is a use the optimiser cannot forward
    // to the argument segment, so in builds where the test survives to that point (the twelve-coefficient polynomial's certified pass did it) the whole 2.2 KB block
    // is copied to scratch by every lane — 0.4 ms per launch — at private offset 0, which the hardware-level compare takes for null: n_frames came out as 1 and the
    // launch's other frames were never written.  profiles/r06_radial_closed_form.txt)
    const int n_frames = clip->n_frames;
#else
    constexpr int n_frames = 1;
    (void)clip;
#endif
    // the planes: four named objects (never an array: nothing may index them dynamically), pointers from the arguments
#if GFW_BAKE
#define GFW_PLANE_INIT(i) GfwYuvPlane PL##i; PL##i.src = A_in.pl[i].src; PL##i.dst = A_in.pl[i].dst; PL##i.src_len = A_in.pl[i].src_len; PL##i.dst_len = A_in.pl[i].dst_len; \
    PL##i.src_stride = GFW_BK_pl##i##_src_stride; PL##i.dst_stride = GFW_BK_pl##i##_dst_stride; PL##i.w = GFW_BK_pl##i##_w; PL##i.h = GFW_BK_pl##i##_h; \
    PL##i.bg[0] = GFW_BK_pl##i##_bg_0; PL##i.bg[1] = GFW_BK_pl##i##_bg_1; PL##i.bg[2] = GFW_BK_pl##i##_bg_2; PL##i.bg[3] = GFW_BK_pl##i##_bg_3; PL##i.limit = GFW_BK_pl##i##_limit; PL##i.fix = GFW_BK_pl##i##_fix; \
    PL##i.org_x = GFW_BK_pl##i##_org_x; PL##i.org_y = GFW_BK_pl##i##_org_y; PL##i.ox32 = GFW_BK_pl##i##_ox32; PL##i.oy32 = GFW_BK_pl##i##_oy32;
    GFW_PLANE_INIT(0) GFW_PLANE_INIT(1) GFW_PLANE_INIT(2) GFW_PLANE_INIT(3)
#undef GFW_PLANE_INIT
#else
    const GfwYuvPlane &PL0 = A_in.pl[0], &PL1 = A_in.pl[1], &PL2 = A_in.pl[2], &PL3 = A_in.pl[3];
    (void)PL3;
#endif
    const float *matrices = A_in.matrices;             // the current frame's table
    // tile = 64 x 4 lanes; each lane owns RB vertically stacked DW x DH luma blocks (+ their chroma sites).
    constexpr int NPX = DW * DH;
    constexpr int QSTEP = NPX > 2 ? 2 : NPX;         // pixels of a lane between two looks at the queue
    constexpr int QCAP = 128 * QSTEP;                // a wave adds at most 64*QSTEP entries between two looks; flushed at half full
    static_assert(RB * NPX <= 64, "slot index must fit the 6 low bits of q_dst");
    // can a projected coordinate be infinite?  Not out of the specialised fisheye projection alone (a*s is bounded by theta_d); every other
    // lens model and any digital lens can produce one, and map_c must then keep it infinite (see map_c)
    constexpr bool INF_COORDS = MODEL != GFW_MODEL_OPENCV_FISHEYE || GFW_BAKED_DIGITAL;
    __shared__ float q_x[FAST1 ? 4 : 1][FAST1 ? QCAP : 1], q_y[FAST1 ? 4 : 1][FAST1 ? QCAP : 1];
    __shared__ unsigned short q_dst[FAST1 ? 4 : 1][FAST1 ? QCAP : 1];           // (owner lane << 6) | slot in s_rows
    // phase-1 rows, already clamped to the matrix table (min(sy, matrix_count - 1): cpu_undistort.rs:482; < 65536: the host keeps larger frames off this path): per luma row
    // of the tile and lane the lane's DW horizontally adjacent pixels side by side — a pair leaves and arrives as one dword
    __shared__ unsigned short s_rows[RB * DH][256][DW];
    auto srow = [&](int slot, int t) -> unsigned short & { return s_rows[slot / DW][t][slot % DW]; };      // slot = r * NPX + j * DW + i (the queue's numbering)
    __shared__ float s_lut[I == 2 ? 1 : 448];                                    // bicubic / Lanczos4 tap table
    __shared__ float4 s_p1[FAST1 ? GFW_CLIP_MAX : 1];                            // per frame of the launch: certificate half-width E, W threshold, (lattice form) rho limit of a node
    // the lattice form of the first pass: a wave's nodes sit on every 8th luma column of its 64 * DW columns (both ends: NXN per row) in its first and its last luma row
    constexpr bool LAT = GFW_P1_LATTICE && FAST1 && (RB * DH > 1);
    constexpr int NXN = (64 * DW) / 8 + 1, HYR = RB * DH - 1;
    __shared__ float s_node[LAT ? 4 : 1][LAT ? 64 : 1];
    const int wave = threadIdx.y, lane = threadIdx.x, tid = wave * 64 + lane;
#if GFW_CK
    unsigned long long ck_acc;                       // the branch-free lane-row's stores: a register pair (an LDS atomic per store cost the kernel 18 %)
#define CKA (&ck_acc)
    // a pair store of the branch-free row sits at an even pixel (luma: pixel cx * DW of its row; interleaved chroma: site cx): inside a word whenever the rows are
    const bool CK_PAIR0 = GFW_BAKE && (PL0.dst_stride % (int)(2 * sizeof(T))) == 0, CK_PAIR1 = GFW_BAKE && (PL1.dst_stride % (int)(2 * sizeof(T))) == 0;
    // ... and where the rows of a plane are whole words apart, a lane's store at a site lands at the SAME place of its word in every row of every tile (its column is
    // tile * 64 + lane): the lane adds the raw values up and shifts each sum once, at the fold.  8- and 16-bit planes: three 32-bit sums — the luma pair's two
    // pixels (ck_l0, ck_l1) and the chroma site's samples (ck_c) — i.e. three v_add_u32 per lane-row of C2 and three registers; a frame's worth of 16-bit samples
    // of one lane cannot overflow them (< 2^15 lane-rows per lane and frame even on the smallest grid: the host sends larger frames through the pass instead).  Float planes: 64-bit sums (ck_l, ck_c64).
    constexpr bool CK_HOT = true;
    const bool CK_INV0 = GFW_BAKE && (PL0.dst_stride & 7) == 0;
    const bool CK_INVC = GFW_BAKE && (PL1.dst_stride & 7) == 0 && (INTERLEAVED_UV || AF(nplanes) < 3 || (PL2.dst_stride & 7) == 0) && (AF(nplanes) < 4 || (PL3.dst_stride & 7) == 0);
    constexpr bool CK_WIDE = sizeof(T) == 4;
    uint32_t ck_l0, ck_l1, ck_c;                     // (zeroed where the tile walk starts: not alive during the set-up above it, whose registers are the kernel's scarcest)
    unsigned long long ck_l, ck_c64;
    auto ck_bits1 = [](uint32_t v) -> uint32_t {         // what a store_value1<T> writes
        if constexpr (is_f32<T>::value && sizeof(T) == 2) return __builtin_bit_cast(uint16_t, (T)__builtin_bit_cast(float, v));
        else return v;
    };
    auto ck_luma = [&](const uint32_t *val) {           // the DW luma values a lane has just stored side by side
        if constexpr (CK_WIDE) { ck_l += (unsigned long long)val[0] + (DW == 2 ? (unsigned long long)val[DW - 1] << 32 : 0ull); }
        else { ck_l0 += ck_bits1(val[0]); if constexpr (DW == 2) ck_l1 += ck_bits1(val[DW - 1]); }
    };
#define CKL (CK_INV0 ? GFW_CK_ELSEWHERE : &ck_acc)
#define CKC (CK_INVC ? GFW_CK_ELSEWHERE : &ck_acc)
    gfw_ck_slot[tid] = 0;                            // (a lane's own slot: nobody else touches it)
    if (lane == 0) gfw_ck_wave[wave] = 0;            // (the wave's own word: ordered before the first fold by that fold's own LDS traffic — one wave, in-order LDS)
    // The wave leaves frame `from` for frame `to` (n_frames at the end) — uniform control flow: it folds its 64 slots (a butterfly through LDS) and lane 0 writes the
    // wave's word of every frame in [from, to) — the sum, then zeros for frames it had no tile of — into the launch's table of partial sums, [frame][workgroup][wave].
    // No atomics: thousands of them on one word serialise across the XCDs (gfw_kernels.hip, gfw_checksum64); gfw_ck_finish adds the table up behind the launch.
    auto ck_flush = [&](int from, int to) {
        unsigned ln = (unsigned)lane, sl = (unsigned)tid, wv = (unsigned)wave;
        asm("" : "+v"(wv));
        asm("" : "+v"(sl));
        asm("" : "+v"(ln));                          // opaque: what the fold derives from the lane — its shifts, its slot — is worked out HERE (hoisted to the top of the
                                                     // kernel these values lived in scratch across all of it, and a kernel that touches scratch starts its 8192 waves
                                                     // slowly enough to cost 8 % of a launch: profiles/r05_c5_checksum.txt)
        const unsigned sh_l = ((ln * (unsigned)(DW * sizeof(T))) & 7u) * 8u;
        const unsigned sh_c = ((ln * (unsigned)((INTERLEAVED_UV ? 2 : 1) * sizeof(T))) & 7u) * 8u;
        const unsigned long long luma = CK_WIDE ? ck_l : (unsigned long long)ck_l0 + ((unsigned long long)ck_l1 << (8 * sizeof(T)));
        const unsigned long long v = gfw_ck_slot[sl] + ck_acc + (luma << sh_l) + ((ck_c64 + ck_c) << sh_c);
        ck_acc = 0; ck_l = 0; ck_c64 = 0; ck_l0 = 0; ck_l1 = 0; ck_c = 0;
        gfw_ck_slot[sl] = 0;
        // the wave's 64 sums into ONE word: 64 LDS additions on the same address — the hardware takes them one after the other, eight times per wave and launch; a
        // butterfly through the lanes' slots needed a dozen registers at a point of the tile walk that has none to spare (they went to scratch: see above)
        GFW_LDS_ADD(&gfw_ck_wave[wv], v);
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        if (ln == 0) {
            const size_t per_frame = (size_t)gridDim.x * 4;
            unsigned long long *o = A.ck_part + (size_t)blockIdx.x * 4 + wv;
            o[(size_t)from * per_frame] = gfw_ck_wave[wv];
            for (int f = from + 1; f < to; ++f) o[(size_t)f * per_frame] = 0;
            gfw_ck_wave[wv] = 0;
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");       // (the word is lane 0's until here)
    };
#else
#define CKA ((unsigned long long *)nullptr)
#define CKL ((unsigned long long *)nullptr)
#define CKC ((unsigned long long *)nullptr)
    const bool CK_PAIR0 = false, CK_PAIR1 = false;
#endif
    if (MODEL == GFW_MODEL_OPENCV_FISHEYE) { gfw_atan_lds_init(tid); gfw_atan_key_lds_init(tid); }
    if (I != 2) {
        for (int i = tid; i < 448; i += 256) s_lut[i] = GFW_COEFFS[i];
        __syncthreads();
    }
    const bool two_pass = AF(matrix_count) > 1 && !GFW_ABL(1) && !AF(fill_bg);
    const bool hrs = AF(hrs) != 0;

    // uniform floats of the pixel loops
    Lens L;
    L.f0 = AFA(f, 0); L.f1 = AFA(f, 1); L.c0 = AFA(c, 0); L.c1 = AFA(c, 1);
    L.k0 = AFA(k, 0); L.k1 = AFA(k, 1); L.k2 = AFA(k, 2); L.k3 = AFA(k, 3);
    L.t2x = AFA(t2, 0); L.t2y = AFA(t2, 1); L.rl2 = AF(r_limit_sq);
    Maps MP;
    MP.mul_lx = AFM(map_lx, mul); MP.mul_ly = AFM(map_ly, mul); MP.mul_cx = AFM(map_cx, mul); MP.mul_cy = AFM(map_cy, mul);
    MP.den_x = AFM(map_lx, den); MP.rcp_x = AFM(map_lx, rcp); MP.den_y = AFM(map_ly, den); MP.rcp_y = AFM(map_ly, rcp);
    float bg_y[N0];
    #pragma unroll
    for (int c = 0; c < N0; ++c) bg_y[c] = PL0.bg[c];
    const float lim_y = PL0.limit;
    float bg_c[2] = {PL1.bg[0], PL1.bg[1]};
    const float lim_u = PL1.limit, bg_v = PL2.bg[0], lim_v = PL2.limit;
    Mid M{0, 0, 0, 0, 0, 0, 0, 0, 0};
    P1 Q{0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    auto load_mid = [&]() {                          // first-pass matrix of the current frame: wave-uniform -> scalar loads
        const float *mid = matrices + (size_t)(AF(matrix_count) >> 1) * GFW_MAT_STRIDE;
        M.m0 = mid[0]; M.m1 = mid[1]; M.m2 = mid[2]; M.m3 = mid[3]; M.m4 = mid[4];
        M.m5 = mid[5]; M.m6 = mid[6]; M.m7 = mid[7]; M.m8 = mid[8];
    };
    // The certificate's half-width per frame of the launch: E = e0 + ew * omega + em * mu with the host's lens-dependent
    // coefficients and two matrix-dependent measures of the rounding error of the linear forms X, Y, W of the frame's mid-row matrix.  With
    // P_X = |ox m0| + |oy m1|, the exact path's X = fl(fl(fl(ox m0) + fl(oy m1)) + m2) is within u (2 P_X + |X|) of the real value and the first
    // pass's fma(oy, m1, fma(ox, m0, m2)) within u (P_X + 2 |X|): mu = 3 max(P_X, P_Y) / W and omega = 3 P_W / W carry the two paths' sum through
    // the division (the |X| terms are in e0).  The P's are maximal at the frame's extreme |ox|, |oy| (literals in a baked build); W >= m8 - P_W
    // everywhere; pixels with W below wmin = max(2^-10, (P_W + |m8|) / 8) are left to the exact path, which keeps omega <= 24 whatever the
    // matrix.  E >= 0.2 px, a NaN anywhere, or an r-limit test that the margin in pass1_fast cannot decide: no pixel of the frame is certified.
    // Thread j of the workgroup evaluates frame j of the launch ONCE, into LDS; a frame change reads two words.  (Evaluated by every wave at
    // every frame change — 35 wave-uniform vector instructions behind the matrix loads — it cost 1.05 us of C2's 46: profiles/r04_ab_certificate.txt.)
    if (FAST1 && two_pass) {
        if (tid < n_frames) {
#if GFW_BAKE
            const float *mid = (tid > 0 ? clip->fr[tid].matrices : matrices) + (size_t)(AF(matrix_count) >> 1) * GFW_MAT_STRIDE;   // (frame 0: the argument block's own table)
#else
            const float *mid = matrices + (size_t)(AF(matrix_count) >> 1) * GFW_MAT_STRIDE;
#endif
            const float m0 = mid[0], m1 = mid[1], m3 = mid[3], m4 = mid[4], m6 = mid[6], m7 = mid[7], m8 = mid[8];
            bool lattice = LAT && !(L.rl2 > 0.0f) && A.p1_lat[5] == 0.0f;            // (p1_lat[5]: the per-pixel form on request — audits of that form, A/B runs)
            // (the lattice's nodes reach up to a tile beyond the frame's last pixel)
            const float ax = fmaxf(fabsf(L.t2x), fabsf((float)(AF(out_w) + (lattice ? 64 * DW : 0)) + L.t2x)), ay = fmaxf(fabsf(L.t2y), fabsf((float)(AF(out_h) + (lattice ? 4 * RB * DH : 0)) + L.t2y));
            const float px = ax * fabsf(m0) + ay * fabsf(m1), py = ax * fabsf(m3) + ay * fabsf(m4), pw = ax * fabsf(m6) + ay * fabsf(m7);
            const float wmin = fmaxf(0.0009765625f, 0.125f * (pw + fabsf(m8)));
            const float wden = fmaxf(m8 - pw, wmin);
            const float rden = gfw_hw_rcp(wden) * 3.003f;                            // (3x; 1 ulp reciprocal and the roundings of the sums above: inside the 0.1 %)
            const float omega = pw * rden, mu = fmaxf(px, py) * rden;
            float E = __builtin_fmaf(A.p1_em, mu, __builtin_fmaf(A.p1_ew, omega, A.p1_eps));
            float rho_lim = A.p1_rho_max;
            const float E_pixel = E;                           // the per-pixel form's half-width (its nodes ARE the pixels)
            if (lattice) {
                // The interpolation's own error: bilinear interpolation of v over a cell of HX x HYR pixels misses it by at most HX^2/8 max|v_xx| + HYR^2/8 max|v_yy|.
                // v = f c S(rho) + c0 with c = b (a for a horizontal shutter), (a, b) = (X, Y) / W, rho = a^2 + b^2.  Where W >= wden and rho <= rho_max:
                //   |a_x| <= (|m0| + rmax |m6|) / wden =: a1x, |b_x| <= (|m3| + rmax |m6|) / wden =: b1x, a_xx = -2 m6 a_x / W, b_xx = -2 m6 b_x / W;  n1 = |(a_x, b_x)|, n2 = |(a_xx, b_xx)|;
                //   |rho_x| <= 2 sqrt(rho) n1, |rho_xx| <= 2 (n1^2 + sqrt(rho) n2)        (Cauchy-Schwarz);
                //   (c S)_xx = c_xx S + 2 c_x S' rho_x + c (S'' rho_x^2 + S' rho_xx),  |c| <= sqrt(rho):
                //   |(c S)_xx| <= c2 S0 + 4 c1 n1 U1 + 4 n1^2 T32 + 2 n1^2 U1 + 2 n2 U2
                // with the host's bounds S0 >= |S|, U1 >= sqrt(rho) |S'|, U2 >= rho |S'|, T32 >= rho^1.5 |S''| over the table's range (gfw_api.hip p1_prepare_table) —
                // the products are bounded together because S'' falls as rho grows: their separate maxima overstate the curvature twentyfold.  The same with m1, m4, m7 for y.
                // The node coordinates' own rounding (ox = fl(lx + t2x)) moves v by u |ox| |v_x|: the last term.  1 % on top for this evaluation's own f32 roundings.
                const float rw = gfw_hw_rcp(wden) * 1.001f, rmax = __builtin_sqrtf(A.p1_rho_max) * 1.0001f;
                const float S0 = A.p1_lat[0], U1 = A.p1_lat[1], U2 = A.p1_lat[2], T32 = A.p1_lat[3];
                const float a1x = (fabsf(m0) + rmax * fabsf(m6)) * rw, b1x = (fabsf(m3) + rmax * fabsf(m6)) * rw;
                const float a1y = (fabsf(m1) + rmax * fabsf(m7)) * rw, b1y = (fabsf(m4) + rmax * fabsf(m7)) * rw;
                const float k6 = 2.0f * fabsf(m6) * rw, k7 = 2.0f * fabsf(m7) * rw;
                const float n1x = __builtin_sqrtf(a1x * a1x + b1x * b1x), n1y = __builtin_sqrtf(a1y * a1y + b1y * b1y);
                const float n2x = k6 * n1x, n2y = k7 * n1y;
                const float c1x = hrs ? a1x : b1x, c1y = hrs ? a1y : b1y;
                const float gxx = k6 * c1x * S0 + 4.0f * c1x * n1x * U1 + n1x * n1x * (4.0f * T32 + 2.0f * U1) + 2.0f * n2x * U2;
                const float gyy = k7 * c1y * S0 + 4.0f * c1y * n1y * U1 + n1y * n1y * (4.0f * T32 + 2.0f * U1) + 2.0f * n2y * U2;
                const float gx = c1x * S0 + 2.0f * U2 * n1x, gy = c1y * S0 + 2.0f * U2 * n1y;                    // |(c S)_x|, |(c S)_y|
                const float hx2 = 8.0f, hy2 = (float)(HYR * HYR) * 0.125f;                                       // HX^2 / 8 (HX = 8), HYR^2 / 8
                E = E + 1.01f * fabsf(AF(p1_f)) * (hx2 * gxx + hy2 * gyy + 5.9604645e-8f * (ax * gx + ay * gy)) + A.p1_lat[4];
                rho_lim = A.p1_rho_max - 1.01f * 2.0f * rmax * (8.0f * n1x + (float)HYR * n1y);                  // |rho| moves by at most this much between a node and any point of its cells
                // The curvature term scales with (8 / f)^2: a 4K frame adds a few thousandths of a pixel, a thumbnail whole pixels.  Beyond GFW_P1_LATTICE_MAX_E (every
                // twelfth pixel undecided) — or a NaN — the frame keeps the per-pixel form and its own, smaller, half-width.
                if (!(E <= GFW_P1_LATTICE_MAX_E) || !(rho_lim > 0.0f)) { lattice = false; E = E_pixel; rho_lim = A.p1_rho_max; }
            }
            bool usable = (E < 0.2f) & (pw + m8 < 3.0e38f) & (rho_lim > 0.0f);        // (a NaN or an infinity among the operands fails a comparison)
            if (L.rl2 > 0.0f) {
                // :139 in pass1_fast: lhs < 0.9999 rhs must imply the exact path's lhs <= rhs.  The two paths' X^2 + Y^2 differ by at most
                // 1.05 u W^2 (2 sqrt2 rmax mu + 12.5 rho_max), their r_limit^2 W by r_limit^2 W u (omega + 8); W <= m8 + P_W; 1e-4 / u = 1677.7
                const float rmax = __builtin_sqrtf(A.p1_rho_max);
                usable = usable & (1.05f * (m8 + pw) * (2.83f * rmax * mu + 12.5f * A.p1_rho_max) + L.rl2 * (omega + 8.0f) <= 1677.0f * L.rl2);
            }
            s_p1[tid] = float4{E, usable ? wmin : __builtin_inff(), rho_lim, lattice ? 1.0f : 0.0f};  // W > inf never holds: every pixel of the frame goes to the exact path
            if (AUDIT && usable) atomicMax(&AF(audit)[6], (unsigned long long)gfw_f2u(E));
        }
        __syncthreads();
    }
    bool p1_lattice = false;                              // the current frame's first pass takes the lattice form
    auto p1_bound = [&](int fi) {
        const float4 q = s_p1[fi];
        Q.eps = gfw_uniform(q.x); Q.wmin = gfw_uniform(q.y); Q.rho_lim = gfw_uniform(q.z); Q.gap = 0.5f - Q.eps;
        p1_lattice = LAT && gfw_uniform(q.w) != 0.0f;
    };
    if (two_pass) {
        load_mid();
        if (FAST1) {
            Q.rho_max = A.p1_rho_max; Q.rho_scale = A.p1_rho_scale; Q.kmax = A.p1_kmax;
            Q.f = AF(p1_f); Q.c = AF(p1_c); Q.lim = (float)(AF(hrs) ? AF(width) : AF(height));
            p1_bound(0);
        }
    }

    // persistent walk over this workgroup's share of the XCD band of tiles
    const int n_tiles = AF(tiles_x) * AF(tiles_y);
    // A frame's tiles form 8 * SUB contiguous sub-bands; XCD x owns sub-bands x, x + 8, x + 16, ... of a frame, rotated by three
    // for every further frame of a clip launch.  An XCD still works inside contiguous runs of tile rows (its L2 sees neighbouring source
    // lines), but cheap regions (the top and bottom of a frame hold most of the out-of-frame pixels) and dear ones no longer land on the
    // same XCDs every frame: one band per XCD measured 54.6 us per C2 frame, rotating it across the clip's frames 51.1.
    const int xcd = (int)blockIdx.x & 7;
    constexpr int SUB = 4;                                // (1: 51.13 us per C2 frame in clip launches / 60.8 frame by frame, 4: 51.02 / 57.9, 8: 50.85)
    const int per_sub = (n_tiles + 8 * SUB - 1) / (8 * SUB);
    const int per_xcd = SUB * per_sub;                    // tile slots of one XCD in one frame (the last sub-bands may run past n_tiles)
    auto xcd_tile = [&](int fi, int r) {                  // slot r of this XCD in frame fi -> tile index (>= n_tiles: none)
        const int h = r / per_sub;
        return ((((xcd + 3 * fi) & 7) + 8 * h) * per_sub) + (r - h * per_sub);
    };
    const int wg_per_xcd = (int)gridDim.x >> 3;
    const int n_slots = per_xcd * n_frames;
#if GFW_TIMELINE
    const unsigned long long tl_ready = wall_clock64();           // set-up done (LDS tables, the frames' certificates, uniforms): the tile walk starts
    unsigned long long tl_p1 = 0, tl_p3 = 0, tl_units = 0;
#if GFW_TIMELINE >= 2
    unsigned long long tl_blk[5] = {0, 0, 0, 0, 0}, tl_mark = 0;
#endif
#endif
#if GFW_PRIO_MODE
    int prio_step = 1;                                // lane-rows per priority level (a GFW_PRIO_SPAN-th of this wave's work)
    auto set_prio = [&](int remaining) {              // s_setprio takes an immediate
        if (remaining >= 3 * prio_step) __builtin_amdgcn_s_setprio(3); else if (remaining >= 2 * prio_step) __builtin_amdgcn_s_setprio(2);
        else if (remaining >= prio_step) __builtin_amdgcn_s_setprio(1); else __builtin_amdgcn_s_setprio(0);
    };
    int tiles_left = 0;
    for (int l = (int)blockIdx.x >> 3; l < n_slots; l += wg_per_xcd) { const int fi = l / per_xcd; if (xcd_tile(fi, l - fi * per_xcd) < n_tiles) ++tiles_left; }
    prio_step = max(1, (tiles_left * RB + GFW_PRIO_SPAN - 1) / GFW_PRIO_SPAN);
#endif
    // l walks (frame, tile slot of this XCD): the frames of a launch are dealt tile by tile like one tall frame.  (Handing the last eighth of
    // a launch's tiles out dynamically, wave by wave from per-XCD counters, levelled the waves' end times — idle wave-slot time 6.3 % -> 3.2 % —
    // and gained nothing: the SIMDs were busy either way, profiles/r03_ab_scheduling.txt.)
    int cur_frame = 0;
#if GFW_CK
    ck_acc = 0; ck_l = 0; ck_c64 = 0; ck_l0 = 0; ck_l1 = 0; ck_c = 0;
#endif
    for (int slot = (int)blockIdx.x >> 3; slot < n_slots; slot += wg_per_xcd) {
        const int fi = n_frames > 1 ? slot / per_xcd : 0;
        const int t = xcd_tile(fi, slot - fi * per_xcd);
        if (t >= n_tiles) continue;                  // the last sub-bands are the short ones
#if GFW_BAKE
        if (fi != cur_frame) {                       // next frame of the launch: its planes and its matrices
#if GFW_CK
            ck_flush(cur_frame, fi);
#endif
            cur_frame = fi;
            PL0.src = clip->fr[fi].src[0]; PL0.dst = clip->fr[fi].dst[0]; PL1.src = clip->fr[fi].src[1]; PL1.dst = clip->fr[fi].dst[1];
            PL2.src = clip->fr[fi].src[2]; PL2.dst = clip->fr[fi].dst[2]; PL3.src = clip->fr[fi].src[3]; PL3.dst = clip->fr[fi].dst[3];
            matrices = clip->fr[fi].matrices;
            if (two_pass) {
                load_mid();
                if (FAST1) p1_bound(fi);
            }
        }
#endif
#if GFW_PRIO_MODE == 1
        set_prio(tiles_left * RB);
#endif
        const int ty = t / AF(tiles_x), tx = t - ty * AF(tiles_x);
        const int cx = tx * 64 + lane;
        const int cy0 = (ty * 4 + wave) * RB;            // first chroma-site row of this lane
        // a frame whose chroma-site grid is whole tiles (4K: 1920 x 2160 sites = 30 x 135 tiles of 64 x 16) needs none of the per-pixel bounds
        // tests; only a baked build knows at compile time (WHOLE folds, the tests below vanish)
        const bool WHOLE = GFW_BAKE && (AF(cw) % 64 == 0) && (AF(ch) % (4 * RB) == 0) && (AF(out_w) == AF(cw) * DW) && (AF(out_h) == AF(ch) * DH);
        const bool lane_ok = WHOLE || cx < AF(cw);

#if GFW_TIMELINE
        const unsigned long long tl_a = __builtin_readcyclecounter();
#endif
        // ---- phase 2 (inside phase 1): the wave resolves its queued pixels exactly, densely packed.  Flushed after the last row, or
        // earlier when the pixels up to the next look (<= 64*QSTEP new entries) could overflow the queue.
        // The queue's length lives in a scalar register (round 4): the lanes that failed their certificate are counted with one wave ballot and take
        // consecutive slots by their rank in it (v_mbcnt) — no LDS atomic with return, no read-back of the length, and the whole enqueue sits behind a
        // uniform branch that the usual pixel (every lane certified) skips.
        unsigned n_q = 0;
        auto flush_queue = [&](bool last) {
            if (last || n_q + 64u * QSTEP > (unsigned)QCAP) {
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");           // the entries the other lanes wrote
                const int row_top = AF(matrix_count) - 1;
                if constexpr (GFW_FASTROW && MODEL == GFW_MODEL_OPENCV_FISHEYE && !AUDIT && !GFW_BAKED_DIGITAL) {
                    // the branch-free projection of phase 3 with the mid-row matrix (round 5): rounds of 64 entries in uniform control flow (its votes ask the wave), the
                    // odd lane — operands outside what the lean sequence is proven for — through pass1_exact itself.  (pass1_exact for all: ~220 issue slots per
                    // round with its divergent branches; this: ~120.)
                    for (unsigned e0 = 0; e0 < n_q; e0 += 64) {
                        const unsigned e = e0 + (unsigned)lane;
                        const bool act = e < n_q;
                        const float ox = q_x[wave][act ? e : 0u], oy = q_y[wave][act ? e : 0u];
                        const float4 ma{M.m0, M.m1, M.m2, M.m3}, mb{M.m4, M.m5, M.m6, M.m7};
                        float pu, pv; bool odd;
                        rd_lean_nobranch<1>(&ox, &oy, &ma, &mb, &M.m8, L, A, &pu, &pv, &odd);
                        const int lim = AF(hrs) ? AF(width) : AF(height);
                        int sy = max(min(round_i32(AF(hrs) ? pu : pv), lim), 0);
                        if (__builtin_expect(gfw_any(odd & act), 0)) { if (odd & act) sy = pass1_exact<MODEL>(ox, oy, M, matrices, L, A); }
                        if (act) { const unsigned d = q_dst[wave][e]; srow((int)(d & 63u), wave * 64 + (int)(d >> 6)) = (unsigned short)min(sy, row_top); }
                    }
                } else {
                    for (unsigned e = lane; e < n_q; e += 64) {
                        const int sy = pass1_exact<MODEL>(q_x[wave][e], q_y[wave][e], M, matrices, L, A);
                        const unsigned d = q_dst[wave][e];
                        srow((int)(d & 63u), wave * 64 + (int)(d >> 6)) = (unsigned short)min(sy, row_top);
                    }
                }
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                n_q = 0;
            }
        };
        // ---- phase 1: rolling-shutter row of every luma pixel of this lane ----------------------------
        if (two_pass && FAST1) {
            // Certified first pass.  Lattice form (round 5; chosen per frame by the certificate's evaluation): the first pass's value v is smooth across a tile and what
            // phase 3 needs of it is clamp(round(v)).  Lanes 0 .. 2 NXN - 1 evaluate v at one node each — columns 0, 8, ..., 64 DW of the wave's span, in its first and its
            // last luma row — every lane then interpolates its own pixels bilinearly between the four nodes around them (a lane's DW pixels never straddle a node column)
            // and certifies the rounded value when no half-integer lies within E of it; E carries the interpolation's error.  A node whose premises fail is NaN, and so is
            // everything interpolated from it: those pixels fail the comparison and are queued for the exact projection like any other undecided pixel.
            // Per-pixel form (rounds 2-4; clips with an r-limit, whose test is per pixel, and frames whose curvature term is too wide): pass1_fast at every pixel.
            const bool lat = LAT && p1_lattice;
            float Tn[DW], Dn[DW];                             // lattice form: the lane's columns — v in the wave's first row, and its change down to the last
            #pragma unroll
            for (int i = 0; i < DW; ++i) { Tn[i] = 0.0f; Dn[i] = 0.0f; }
            if (LAT && lat) {
                int ln = lane, wv = wave;
                asm("" : "+v"(ln));               // opaque: the nodes' offsets and LDS addresses are a handful of instructions per tile; hoisted out of the tile walk they
                asm("" : "+v"(wv));               // lived in registers for the whole kernel, and two of them spilled
                const int niy = ln >= NXN ? 1 : 0, nix = ln - niy * NXN;                      // (lanes beyond 2 NXN evaluate a point nobody reads)
                const float nox = (float)(tx * (64 * DW) + nix * 8) + L.t2x, noy = (float)(cy0 * DH + niy * HYR) + L.t2y;
                s_node[wv][ln] = pass1_node(nox, noy, M, Q, A.p1_table, hrs);
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                const int ci = (ln * DW) >> 3;
                const float v00 = s_node[wv][ci], v01 = s_node[wv][ci + 1], v10 = s_node[wv][NXN + ci], v11 = s_node[wv][NXN + ci + 1];
                const float dT = v01 - v00, dB = v11 - v10;
                #pragma unroll
                for (int i = 0; i < DW; ++i) {
                    const float txi = (float)((ln * DW + i) & 7) * 0.125f;
                    Tn[i] = __builtin_fmaf(txi, dT, v00);
                    Dn[i] = __builtin_fmaf(txi, dB, v10) - Tn[i];
                }
            }
            #pragma unroll 1
            for (int q = 0; q < RB * DH; ++q) {                // the lane's luma rows, top to bottom
                const int r = q / DH, j = q % DH;
                const int ly = cy0 * DH + q;
                const float oy = (float)ly + L.t2y;
                // every lane evaluates (a dead lane's coordinates are as good as any); "certified" travels as a wave mask — the compare's own result, in scalar registers
                GfwVote good[DW]; int sy[DW]; float v_fast[DW];
                if (LAT && lat) {
                    const float ty = (float)q * (1.0f / (float)(HYR > 0 ? HYR : 1));
                    #pragma unroll
                    for (int i = 0; i < DW; ++i) {
                        v_fast[i] = __builtin_fmaf(ty, Dn[i], Tn[i]);
                        const float n = rintf(v_fast[i]);
                        good[i] = gfw_lanes(fabsf(v_fast[i] - n) < Q.gap);           // (NaN: not certified)
                        sy[i] = max(min(gfw_f2i(n), min((int)Q.lim, AF(matrix_count) - 1)), 0);      // (both upper clamps at once: :469 and :482)
                    }
                } else {
                    #pragma unroll
                    for (int i = 0; i < DW; ++i) {
                        const float ox = (float)(cx * DW + i) + L.t2x;
                        const float ax = __builtin_fmaf(ox, M.m0, M.m2), ay = __builtin_fmaf(ox, M.m3, M.m5), aw = __builtin_fmaf(ox, M.m6, M.m8);
                        good[i] = gfw_lanes(pass1_fast(ax, ay, aw, oy, M, Q, A.p1_table, hrs, L.rl2, sy[i], v_fast[i], AUDIT ? AF(audit) : nullptr));
                    }
                }
                #pragma unroll
                for (int i = 0; i < DW; ++i) {
                    const int lx = cx * DW + i;
                    const bool live = WHOLE || (lane_ok && lx < AF(out_w) && ly < AF(out_h));
                    if (!WHOLE) good[i] = good[i] | gfw_lanes(!live);
                    sy[i] = live ? ((LAT && lat) ? sy[i] : min(sy[i], AF(matrix_count) - 1)) : 0;
                    if constexpr (DW != 2) s_rows[q][tid][i] = (unsigned short)sy[i];
                    if (AUDIT && gfw_vote_lane(good[i], lane) && live) {           // audit: every certificate is checked
                        const float ox = (float)lx + L.t2x;
                        atomicAdd(&AF(audit)[0], 1ull);
                        if (min(pass1_exact<MODEL>(ox, oy, M, matrices, L, A), AF(matrix_count) - 1) != sy[i]) atomicAdd(&AF(audit)[1], 1ull);
                        const GfwPt ex = rd<MODEL>(ox, oy, float4{M.m0, M.m1, M.m2, M.m3}, float4{M.m4, M.m5, M.m6, M.m7}, M.m8, matrices + (size_t)(AF(matrix_count) / 2) * GFW_MAT_STRIDE + 8, L, A);
                        if (ex.ok) atomicMax(&AF(audit)[4], (unsigned long long)gfw_f2u(fabsf((hrs ? ex.x : ex.y) - v_fast[i])));
                    }
                }
                if constexpr (DW == 2) *reinterpret_cast<uint32_t *>(&s_rows[q][tid][0]) = (uint32_t)sy[0] | ((uint32_t)sy[1] << 16);       // the pair as one dword
                unsigned long long undecided[DW], any_und = 0ull;
                #pragma unroll
                for (int i = 0; i < DW; ++i) { undecided[i] = gfw_vote_failed(good[i]); any_und |= undecided[i]; }
                if (any_und) {                                 // uniform: the usual row certifies every pixel
                    #pragma unroll
                    for (int i = 0; i < DW; ++i) {
                        if (!gfw_vote_lane(good[i], lane)) {
                            const unsigned slot = n_q + __builtin_amdgcn_mbcnt_hi((unsigned)(undecided[i] >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)undecided[i], 0u));   // < QCAP: flushed before it can fill
                            q_x[wave][slot] = (float)(cx * DW + i) + L.t2x; q_y[wave][slot] = oy;
                            q_dst[wave][slot] = (unsigned short)((lane << 6) | (r * NPX + j * DW + i));
                            if (AUDIT) atomicAdd(&AF(audit)[2], 1ull);
                        }
                        n_q += (unsigned)__popcll(undecided[i]);
                    }
                }
                flush_queue(q == RB * DH - 1);                 // (a look at the queue per row: at most 64 DW new entries since the last)
            }
        } else if (two_pass) {
            #pragma unroll 1
            for (int r = 0; r < RB; ++r) {
                #pragma unroll (NPX <= 2 ? NPX : 1)
                for (int k = 0; k < NPX; ++k) {
                    const int i = k % DW, j = k / DW;
                    const int lx = cx * DW + i, ly = (cy0 + r) * DH + j;
                    const bool live = WHOLE || (lane_ok && lx < AF(out_w) && ly < AF(out_h));
                    int sy = 0;
                    if (live) {
                        float ox = (float)lx + L.t2x, oy = (float)ly + L.t2y;
                        if (MODEL != GFW_MODEL_OPENCV_FISHEYE && (AF(extras) & 8)) gfw_lens_correction_blend<(MODEL == GFW_MODEL_GENERIC_EXTRA ? -1 : MODEL)>(ox, oy, A.kp, A.common, AF(model), GFW_CLIP_DIGITAL);   // :429-460
                        sy = pass1_exact<MODEL>(ox, oy, M, matrices, L, A);
                    }
                    srow(r * NPX + k, tid) = (unsigned short)(live ? min(sy, AF(matrix_count) - 1) : 0);
                }
            }
        }

#if GFW_TIMELINE
        const unsigned long long tl_b = __builtin_readcyclecounter();
#endif
        // ---- phase 3: exact projection with the row's own matrix, then taps ---------------------------
        if (lane_ok) {
            #pragma unroll 1                       // (2 or 4 lane-rows per iteration: 45.9 = 45.8 us)
            for (int r = 0; r < RB; ++r) {
                const int cy = cy0 + r;
                if (!WHOLE && cy >= AF(ch)) break;
                if (AF(fill_bg)) {
                    // FILL_WITH_BACKGROUND (cpu_undistort.rs:558-561; the render loop sets it for frames outside the trim ranges): `*pix_out = bg_t` for every
                    // pixel of the plane — no projection, no taps
                    #pragma unroll
                    for (int k = 0; k < NPX; ++k) {
                        const int lx = cx * DW + k % DW, ly = cy * DH + k / DW;
                        if (WHOLE || (lx < AF(out_w) && ly < AF(out_h))) store_px<T, N0>(PL0.dst, row_off(ly, PL0.dst_stride) + lx * (int)(N0 * sizeof(T)), bg_y);
                    }
                    if (AF(nplanes) > 1) {
                        if (INTERLEAVED_UV) store_px<T, 2>(PL1.dst, row_off(cy, PL1.dst_stride) + cx * (int)(2 * sizeof(T)), bg_c);
                        else {
                            const int doff = row_off(cy, PL1.dst_stride) + cx * (int)sizeof(T);
                            store_px<T, 1>(PL1.dst, doff, PL1.bg);
                            if (AF(nplanes) > 2) store_px<T, 1>(PL2.dst, doff, PL2.bg);
                            if (AF(nplanes) > 3) store_px<T, 1>(PL3.dst, doff, PL3.bg);
                        }
                    }
                    continue;
                }
                float u0 = 0.0f, v0 = 0.0f, lu0 = 0.0f, lv0 = 0.0f; bool ok0 = false;
                GfwVote okm0 = GFW_VOTE_ALL;              // the branch-free row's form of ok0
                bool row_done = false;                    // GFW_ROW_CLUSTER: luma and chroma of this lane-row already written
                // The branch-free row (round 4; specialised fisheye, bilinear, single-channel luma): a lane's DW pixels of one line are projected with
                // rd_lean_nobranch, mapped and binned without a divergent branch; the wave is asked ONCE per stage — `__any(rare)` sends the odd lanes
                // through rd<> itself, `__all(interior)` picks between the branch-free taps (pair stored as one word) and sample_store2.
                constexpr bool FASTROW = GFW_FASTROW && MODEL == GFW_MODEL_OPENCV_FISHEYE && N0 == 1 && !AUDIT && !GFW_BAKED_DIGITAL;      // (packed RGBAf through this row measured 69.8 against 64.0 us per C4 frame: profiles/r05_ab_c4_fastrow.txt)
                // (GFW_ABL: 0 in this library; GFW_TESTING builds: 1 no first pass, 2 no luma taps (the store stays), 4 no chroma, 8 no projection, 16 no luma store,
                //  32 every pixel's matrix = the mid row's, 64 the second pixel's taps by DPP from the neighbouring lane)
                const bool fastrow = FASTROW && (GFW_BAKE || !GFW_ABL(~0)) && !AF(fix_range) && !AF(hstretch_div) && !AF(vstretch_div) && !AF(rot_on);       // (a stretched or rotated clip: the per-pixel path)
                if (fastrow) {
                    #pragma unroll (NPX <= 2 ? DH : 1)
                    for (int j = 0; j < DH; ++j) {
                        const int ly = cy * DH + j;
                        GFW_TLB_START();
                        float pu[DW], pv[DW]; bool odd[DW]; bool any_odd = false;
                        GfwVote okp[DW];                 // which lanes hold a valid point: a wave mask in scalar registers (a per-lane bool would live in a VGPR across the vote below)
                        {
                            float ox[DW], oy[DW], m8[DW]; float4 ma[DW], mb[DW];
                            uint32_t rows2 = 0u;                  // the pair's two rows, one dword (already clamped to the table)
                            if constexpr (DW == 2) if (two_pass) rows2 = *reinterpret_cast<const uint32_t *>(&s_rows[r * DH + j][tid][0]);
                            int row[DW];
                            #pragma unroll
                            for (int i = 0; i < DW; ++i) {
                                const int lx = cx * DW + i;
                                // (translation2d == +0 in a baked build: x + (+0) is x for every x >= +0, and (float)(lx + 1) is (float)lx + 1 below 2^24 — a conversion and three adds per pair)
                                if (GFW_BAKE && gfw_f2u(L.t2x) == 0u) ox[i] = (i == 0) ? (float)lx : ox[0] + (float)i; else ox[i] = (float)lx + L.t2x;
                                if (GFW_BAKE && gfw_f2u(L.t2y) == 0u) oy[i] = (i == 0) ? (float)ly : oy[0]; else oy[i] = (float)ly + L.t2y;
                                row[i] = two_pass ? (DW == 2 ? (int)(i == 0 ? (rows2 & 0xffffu) : (rows2 >> 16)) : (int)s_rows[r * DH + j][tid][i])
                                                  : min(default_row<MODEL>(ox[i], oy[i], A), AF(matrix_count) - 1);
                            }
                            // (the rows of a lane-row through a 16-row LDS window — one cooperative dwordx4 instead of these six fetches — measured 42.3 = 42.3 us per C2
                            //  frame with 27 % fewer L1 accesses: profiles/r05_c2_memory_path.txt; not kept)
                            #pragma unroll
                            for (int i = 0; i < DW; ++i) {
                                const float *m = reinterpret_cast<const float *>(reinterpret_cast<const uint8_t *>(matrices) + (uint32_t)row[i] * (uint32_t)(GFW_MAT_STRIDE * sizeof(float)));
                                ma[i] = *reinterpret_cast<const float4 *>(m); mb[i] = *reinterpret_cast<const float4 *>(m + 4); m8[i] = m[8];
                            }
                            if (GFW_BAKE && GFW_ABL(32)) {
                                #pragma unroll
                                for (int i = 0; i < DW; ++i) { ma[i] = float4{M.m0, M.m1, M.m2, M.m3}; mb[i] = float4{M.m4, M.m5, M.m6, M.m7}; m8[i] = M.m8; }
                            }
#if GFW_TIMELINE >= 2
                            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                            GFW_TLB(0);
#endif
                            rd_lean_nobranch<DW>(ox, oy, ma, mb, m8, L, A, pu, pv, odd);         // (the pair side by side or one after the other: 46.2 = 46.4 us)
                            GFW_TLB(1);
                            if (GFW_BAKE && GFW_ABL(8)) {
                                #pragma unroll
                                for (int i = 0; i < DW; ++i) { pu[i] = ox[i] * 0.5f; pv[i] = oy[i] * 0.5f; odd[i] = false; }
                            }
                            #pragma unroll
                            for (int i = 0; i < DW; ++i) { okp[i] = GFW_VOTE_ALL; any_odd = any_odd | odd[i]; }
                        }
                        if (__builtin_expect(gfw_any(any_odd), 0)) {
                            #pragma unroll
                            for (int i = 0; i < DW; ++i) {
                                bool ok_i = true;
                                if (odd[i]) {
                                    const int lx = cx * DW + i;
                                    const float ox = (float)lx + L.t2x, oy = (float)ly + L.t2y;
                                    const int row = two_pass ? (int)s_rows[r * DH + j][tid][i] : min(default_row<MODEL>(ox, oy, A), AF(matrix_count) - 1);
                                    const GfwPt p = rd_row<MODEL>(ox, oy, row, matrices, L, A);
                                    pu[i] = p.x; pv[i] = p.y; ok_i = p.ok;
                                }
                                okp[i] = gfw_vote_select(okp[i], gfw_lanes(odd[i]), gfw_lanes(ok_i));
                            }
                        }
                        int bx[DW], by[DW]; GfwVote interior = GFW_VOTE_ALL;      // (bx, by) = round(32 u), round(32 v): all a sample's bins and weights derive from them
                        #pragma unroll
                        for (int i = 0; i < DW; ++i) {
                            const int lx = cx * DW + i;
                            if (AF(background_mode) == 1) {                               // cpu_undistort.rs:495-509 (edge repeat / edge mirror), as below
                                const float width_f = (float)AF(width), height_f = (float)AF(height);
                                pu[i] = fminf(fmaxf(pu[i], 3.0f), width_f - 3.0f);
                                pv[i] = fminf(fmaxf(pv[i], 3.0f), height_f - 3.0f);
                            } else if (AF(background_mode) == 2) {
                                const float width_f = (float)AF(width), height_f = (float)AF(height);
                                const float rx = roundf(pu[i]), ry = roundf(pv[i]);
                                const float width3 = width_f - 3.0f, height3 = height_f - 3.0f;
                                if (rx > width3)  pu[i] = width3  - (rx - width3);
                                if (rx < 3.0f)    pu[i] = 3.0f + width_f - (width3  + rx);
                                if (ry > height3) pv[i] = height3 - (ry - height3);
                                if (ry < 3.0f)    pv[i] = 3.0f + height_f - (height3 + ry);
                            }
                            const float lu = map_c<INF_COORDS>(pu[i], MP.mul_lx, MP.den_x, MP.rcp_x), lv = map_c<INF_COORDS>(pv[i], MP.mul_ly, MP.den_y, MP.rcp_y);   // :511-514
                            if (j == 0 && i == 0) { u0 = pu[0]; v0 = pv[0]; okm0 = okp[0]; lu0 = lu; lv0 = lv; }
                            bx[i] = bin_x<I>(lu, PL0); by[i] = bin_y<I>(lv, PL0);
                            const bool live = WHOLE || (lx < AF(out_w) && ly < AF(out_h));
                            if constexpr (I == 2) interior = interior & okp[i] & gfw_lanes(live) & gfw_lanes((unsigned)(bx[i] >> 5) < (unsigned)(PL0.w - 1)) & gfw_lanes((unsigned)(by[i] >> 5) < (unsigned)(PL0.h - 1));
                            else interior = interior & okp[i] & gfw_lanes(live) & lut_interior<T, I>(bx[i], by[i], PL0.w, PL0.h);
                        }
                        if constexpr (GFW_ROW_CLUSTER && GFW_BAKE && MODEL == GFW_MODEL_OPENCV_FISHEYE && I == 2 && DH == 1 && !INTERLEAVED_UV && !is_f32<T>::value) {
                            // the chroma site's bins and its interior vote BEFORE any tap: with both votes in hand the row's eight fetches leave together
                            if (AF(nplanes) == 3 && !GFW_ABL(4)) {
                                const float ccu = chroma_from_luma<INF_COORDS>(lu0, u0, MP.mul_cx, MP.mul_lx, MP.den_x, MP.rcp_x);
                                const float ccv = chroma_from_luma<INF_COORDS>(lv0, v0, MP.mul_cy, MP.mul_ly, MP.den_y, MP.rcp_y);
                                const Bins2 bc = bins2_of(bin_x<2>(ccu, PL1), bin_y<2>(ccv, PL1));
                                const GfwVote both = interior & okm0 & gfw_lanes((unsigned)bc.sx < (unsigned)(PL1.w - 1)) & gfw_lanes((unsigned)bc.sy < (unsigned)(PL1.h - 1));
                                if (__builtin_expect(gfw_all_lanes(both), 1)) {
                                    uint32_t val[DW];
                                    #pragma unroll
                                    for (int i = 0; i < DW; ++i) val[i] = inside_value1<T>(PL0.src, PL0.src_stride, bins2_of(bx[i], by[i]), bg_y, lim_y);
                                    const uint32_t vu = inside_value1<T>(PL1.src, PL1.src_stride, bc, &bg_c[0], lim_u);
                                    const uint32_t vv = inside_value1<T>(PL2.src, PL1.src_stride, bc, &bg_v, lim_v);
                                    const int doff = row_off(ly, PL0.dst_stride) + (cx * DW) * (int)sizeof(T);
                                    if constexpr (DW == 2) store_pair1<T>(PL0.dst, doff, val[0], val[1], CKA, CK_PAIR0);
                                    else store_value1<T>(PL0.dst, doff, val[0], CKA);
                                    const int cdoff = row_off(cy, PL1.dst_stride) + cx * (int)sizeof(T);
                                    store_value1<T>(PL1.dst, cdoff, vu, CKA); store_value1<T>(PL2.dst, cdoff, vv, CKA);
                                    row_done = true;
                                }
                            }
                        }
                        if (row_done) {}
                        else if (__builtin_expect(gfw_all_lanes(interior), 1)) {
                            uint32_t val[DW];
                            #pragma unroll
                            for (int i = 0; i < DW; ++i) {
                                if (GFW_BAKE && GFW_ABL(2)) val[i] = (uint32_t)(bx[i] ^ by[i]) & 0xffu;
                                else if (GFW_BAKE && GFW_ABL(64) && I == 2 && sizeof(T) == 2 && i == 1) {
// [GFW-TESTING-BEGIN]
#if GFW_TESTING && !defined(GFW_HOST_INTERPRETER)
                                    // ablation 64 (wrong output by design): the upper bound of the north-star's "wavefront-level gather for the bilinear tap" — the pair's
                                    // SECOND pixel takes its two row dwords from the neighbouring lane's first pixel through DPP (row_shr:1) instead of fetching them; what real
                                    // sharing could save at most, were the neighbour's rows and columns always the right ones (they are not: DESIGN.md section 4)
                                    const Bins2 b0 = bins2_of(bx[0], by[0]), b1 = bins2_of(bx[1], by[1]);
                                    const int o0 = row_off(b0.sy, PL0.src_stride) + b0.sx * 2;
                                    const uint32_t d0 = *reinterpret_cast<const uint32_t __attribute__((aligned(2))) *>(PL0.src + (uint32_t)o0);
                                    const uint32_t d1 = *reinterpret_cast<const uint32_t __attribute__((aligned(2))) *>(PL0.src + (uint32_t)o0 + (uint32_t)PL0.src_stride);
                                    auto blend = [&](uint32_t r0, uint32_t r1, const Bins2 &bb) {
                                        const float xs0 = __builtin_fmaf((float)(r0 >> 16), bb.cx1, (float)(r0 & 0xffffu) * bb.cx0);
                                        const float xs1 = __builtin_fmaf((float)(r1 >> 16), bb.cx1, (float)(r1 & 0xffffu) * bb.cx0);
                                        return gfw_f2u_trunc(min_limit(xs0 * bb.cy0 + xs1 * bb.cy1, lim_y));
                                    };
                                    val[0] = blend(d0, d1, b0);
                                    val[1] = blend((uint32_t)__builtin_amdgcn_update_dpp(0, (int)d0, 0x111, 0xf, 0xf, false), (uint32_t)__builtin_amdgcn_update_dpp(0, (int)d1, 0x111, 0xf, 0xf, false), b1);
#endif
// [GFW-TESTING-END]
                                }
                                else if constexpr (I == 2) val[i] = inside_value1<T>(PL0.src, PL0.src_stride, bins2_of(bx[i], by[i]), bg_y, lim_y);
                                else val[i] = inside_value1_lut<T, I>(PL0.src, PL0.src_stride, bx[i], by[i], bg_y, lim_y, s_lut);
                            }
                            const int doff = row_off(ly, PL0.dst_stride) + (cx * DW) * (int)sizeof(T);
                            if (GFW_BAKE && GFW_ABL(16)) { if (lane == 99 && val[0] == 0x12345u) PL0.dst[0] = 1; }
                            else if constexpr (DW == 2) store_pair1<T>(PL0.dst, doff, val[0], val[1], CKL, CK_PAIR0);
                            else store_value1<T>(PL0.dst, doff, val[0], CKL);
#if GFW_CK
                            if (CK_INV0 && CK_HOT) ck_luma(val);
#endif
                        } else {
                            #pragma unroll
                            for (int i = 0; i < DW; ++i) {
                                const int lx = cx * DW + i;
                                if (WHOLE || (lx < AF(out_w) && ly < AF(out_h))) {
                                    if constexpr (I == 2) sample_store2_bins<T, N0>(bx[i], by[i], gfw_vote_lane(okp[i], lane), PL0, bg_y, lim_y, lx, ly, nullptr);
                                    else sample_store_bins<T, N0, I>(bx[i], by[i], gfw_vote_lane(okp[i], lane), PL0, bg_y, lim_y, lx, ly, s_lut);
                                }
                            }
                        }
                        GFW_TLB(2);
                    }
                } else {
                #pragma unroll (NPX <= 2 ? NPX : 1)
                for (int k = 0; k < NPX; ++k) {
                    const int i = k % DW, j = k / DW;
                    const int lx = cx * DW + i, ly = cy * DH + j;
                    if (!WHOLE && (lx >= AF(out_w) || ly >= AF(out_h))) continue;
                    float ox = (float)lx + L.t2x, oy = (float)ly + L.t2y;
                    if (MODEL != GFW_MODEL_OPENCV_FISHEYE && (AF(extras) & 8)) gfw_lens_correction_blend<(MODEL == GFW_MODEL_GENERIC_EXTRA ? -1 : MODEL)>(ox, oy, A.kp, A.common, AF(model), GFW_CLIP_DIGITAL);       // :429-460
                    const int sy = two_pass ? (int)srow(r * NPX + k, tid) : default_row<MODEL>(ox, oy, A);      // (two_pass: already clamped to the table; the min below is then idle)
                    GfwPt p;
                    if (GFW_ABL(8)) { p.x = ox * 0.5f; p.y = oy * 0.5f; p.ok = true; }              // timing ablation only
                    else {
                        const int row = min(sy, AF(matrix_count) - 1);
                        if (AUDIT && (unsigned)row >= (unsigned)AF(matrix_count)) atomicAdd(&AF(audit)[5], 1ull);
                        p = rd_row<MODEL>(ox, oy, row, matrices, L, A);
                    }
                    if (AF(rot_on) && p.ok) {                                                          // input_rotation (cpu_undistort.rs:485-491): rotate_point(uv, rot, size/2, frame_size/2),
                        const float ox_ = (float)AF(width) / 2.0f, oy_ = (float)AF(height) / 2.0f;      // cos / sin from the host libm as the reference's f32::cos / sin
                        const float o2x = A.common.frame_w / 2.0f, o2y = A.common.frame_h / 2.0f;
                        const float rx = A.common.rot_cos * (p.x - ox_) - A.common.rot_sin * (p.y - oy_) + o2x;
                        const float ry = A.common.rot_sin * (p.x - ox_) + A.common.rot_cos * (p.y - oy_) + o2y;
                        p.x = rx; p.y = ry;
                    }
                    if ((AF(background_mode) == 1 || AF(background_mode) == 2) && p.ok) {                      // cpu_undistort.rs:495-509 (edge repeat / edge mirror)
                        const float width_f = (float)AF(width), height_f = (float)AF(height);
                        if (AF(background_mode) == 1) {
                            p.x = fminf(fmaxf(p.x, 3.0f), width_f - 3.0f);
                            p.y = fminf(fmaxf(p.y, 3.0f), height_f - 3.0f);
                        } else {
                            const float rx = roundf(p.x), ry = roundf(p.y);
                            const float width3 = width_f - 3.0f, height3 = height_f - 3.0f;
                            if (rx > width3)  p.x = width3  - (rx - width3);
                            if (rx < 3.0f)    p.x = 3.0f + width_f - (width3  + rx);
                            if (ry > height3) p.y = height3 - (ry - height3);
                            if (ry < 3.0f)    p.y = 3.0f + height_f - (height3 + ry);
                        }
                    }
                    if (k == 0) { u0 = p.x; v0 = p.y; ok0 = p.ok; }
                    if (MODEL == GFW_MODEL_GENERIC_EXTRA && (AF(extras) & 16) && p.ok) {          // background mode 3: two samples, blended (:576-613)
                        feather_store<T, N0, I, true>(p.x, p.y, feather_of(p.x, p.y, A), PL0, bg_y, lim_y, MP.mul_lx, MP.mul_ly, MP, lx, ly, s_lut);
                        continue;
                    }
                    const float lu = map_c<INF_COORDS>(p.x, MP.mul_lx, MP.den_x, MP.rcp_x), lv = map_c<INF_COORDS>(p.y, MP.mul_ly, MP.den_y, MP.rcp_y);   // cpu_undistort.rs:511-514
                    if (GFW_ABL(2)) { if (lane == 99) PL0.dst[0] = (uint8_t)(lu + lv); continue; }  // timing ablation only
                    if (k == 0) { lu0 = lu; lv0 = lv; }
                    if (I == 2) sample_store2<T, N0>(lu, lv, p.ok, PL0, bg_y, lim_y, lx, ly, AUDIT ? AF(audit) : nullptr);
                    else sample_store<T, N0, I>(lu, lv, p.ok, PL0, bg_y, lim_y, lx, ly, s_lut);
                }
                }
                if (MODEL == GFW_MODEL_GENERIC_EXTRA && (AF(extras) & 16) && ok0 && AF(nplanes) > 1) {  // background mode 3 for the chroma site
                    const Feather f = feather_of(u0, v0, A);
                    if (INTERLEAVED_UV) feather_store<T, 2, I, true>(u0, v0, f, PL1, bg_c, lim_u, MP.mul_cx, MP.mul_cy, MP, cx, cy, s_lut);
                    else {
                        // the named planes, not A_in.pl[]: in a clip launch they carry the CURRENT frame's pointers (the argument block's own are frame 0's)
                        feather_store<T, 1, I, true>(u0, v0, f, PL1, PL1.bg, PL1.limit, MP.mul_cx, MP.mul_cy, MP, cx, cy, s_lut);
                        if (AF(nplanes) > 2) feather_store<T, 1, I, true>(u0, v0, f, PL2, PL2.bg, PL2.limit, MP.mul_cx, MP.mul_cy, MP, cx, cy, s_lut);
                        if (AF(nplanes) > 3) feather_store<T, 1, I, true>(u0, v0, f, PL3, PL3.bg, PL3.limit, MP.mul_cx, MP.mul_cy, MP, cx, cy, s_lut);
                    }
                } else
                if (row_done) {}
                else if (AF(nplanes) > 1 && !GFW_ABL(4)) {
                    float cu, cv;
                    if (GFW_BAKE && MODEL == GFW_MODEL_OPENCV_FISHEYE) {
                        // the chroma site's coordinate from the luma pixel's that shares it (chroma_from_luma): nothing for 4:2:2's rows, one multiply for a halved axis
                        cu = chroma_from_luma<INF_COORDS>(lu0, u0, MP.mul_cx, MP.mul_lx, MP.den_x, MP.rcp_x);
                        cv = chroma_from_luma<INF_COORDS>(lv0, v0, MP.mul_cy, MP.mul_ly, MP.den_y, MP.rcp_y);
                    } else {
                        cu = map_c<INF_COORDS>(u0, MP.mul_cx, MP.den_x, MP.rcp_x);
                        cv = map_c<INF_COORDS>(v0, MP.mul_cy, MP.den_y, MP.rcp_y);
                    }
                    bool chroma_done = false;
                    if constexpr (FASTROW && I == 2 && !is_f32<T>::value) if (fastrow && (INTERLEAVED_UV || AF(nplanes) == 3)) {
                        // the chroma site the same way: one question to the wave, then the branch-free interior taps of both chroma samples
                        const Bins2 bc = bins2_of(bin_x<2>(cu, PL1), bin_y<2>(cv, PL1));
                        if (__builtin_expect(gfw_all_lanes(okm0 & gfw_lanes((unsigned)bc.sx < (unsigned)(PL1.w - 1)) & gfw_lanes((unsigned)bc.sy < (unsigned)(PL1.h - 1))), 1)) {
                            if constexpr (INTERLEAVED_UV) {
                                const int off0 = row_off(bc.sy, PL1.src_stride) + bc.sx * (int)(2 * sizeof(T));
                                const int doff = row_off(cy, PL1.dst_stride) + cx * (int)(2 * sizeof(T));
                                if constexpr (sizeof(T) == 1) {
                                    typedef HotTap<T, true> Tap;
                                    const auto r0 = Tap::load(PL1.src, (uint32_t)off0), r1 = Tap::load(PL1.src, (uint32_t)off0 + (uint32_t)PL1.src_stride);
                                    const uint32_t w = Tap::wpack(bc.kx);
                                    uint32_t ua, va, ub, vb;
                                    Tap::dot(r0, w, ua, va); Tap::dot(r1, w, ub, vb);
                                    store_pair1<T>(PL1.dst, doff, hot_blend(ua, ub, bc.ky, lim_u), hot_blend(va, vb, bc.ky, lim_u), CKA, CK_PAIR1);      // (8-bit pairs: through the lane's general sum)
                                } else {
                                    float o[2];
                                    taps_inside2<T, 2>(PL1.src, off0, PL1.src_stride, bc, lim_u, o);
                                    const bool sat = px_needs_sat<T>(bg_c, 2, lim_u);
                                    const uint32_t p0 = sat ? gfw_f2u_sat(o[0], 65535.0f) : gfw_f2u_trunc(o[0]), p1 = sat ? gfw_f2u_sat(o[1], 65535.0f) : gfw_f2u_trunc(o[1]);
                                    store_pair1<T>(PL1.dst, doff, p0, p1, CKC, CK_PAIR1);
#if GFW_CK
                                    if (CK_INVC) ck_c64 += (unsigned long long)p0 + ((unsigned long long)p1 << (8 * sizeof(T)));      // (interleaved 16-bit chroma: one 64-bit sum)
#endif
                                }
                            } else {
                                const uint32_t vu = inside_value1<T>(PL1.src, PL1.src_stride, bc, &bg_c[0], lim_u);
                                const uint32_t vv = inside_value1<T>(PL2.src, PL1.src_stride, bc, &bg_v, lim_v);
                                const int doff = row_off(cy, PL1.dst_stride) + cx * (int)sizeof(T);
                                store_value1<T>(PL1.dst, doff, vu, CKC); store_value1<T>(PL2.dst, doff, vv, CKC);
#if GFW_CK
                                if (CK_INVC && CK_HOT) { if constexpr (CK_WIDE) ck_c64 += (unsigned long long)vu + vv; else ck_c += ck_bits1(vu) + ck_bits1(vv); }
#endif
                            }
                            chroma_done = true;
                        }
                    }
                    if constexpr (FASTROW && I == 2 && is_f32<T>::value && N0 == 1 && !INTERLEAVED_UV) if (fastrow && AF(nplanes) >= 2) {
                        // planar float frames (the EXR route: G, B, R, A planes of one geometry — round 5): one vote, then every further plane's interior taps with one set of bins
                        const Bins2 bc = bins2_of(bin_x<2>(cu, PL1), bin_y<2>(cv, PL1));
                        if (__builtin_expect(gfw_all_lanes(okm0 & gfw_lanes((unsigned)bc.sx < (unsigned)(PL1.w - 1)) & gfw_lanes((unsigned)bc.sy < (unsigned)(PL1.h - 1))), 1)) {
                            const uint32_t va = inside_value1<T>(PL1.src, PL1.src_stride, bc, PL1.bg, PL1.limit);
                            const uint32_t vb = AF(nplanes) > 2 ? inside_value1<T>(PL2.src, PL1.src_stride, bc, PL2.bg, PL2.limit) : 0u;
                            const uint32_t vc = AF(nplanes) > 3 ? inside_value1<T>(PL3.src, PL1.src_stride, bc, PL3.bg, PL3.limit) : 0u;
                            const int doff = row_off(cy, PL1.dst_stride) + cx * (int)sizeof(T);
                            store_value1<T>(PL1.dst, doff, va, CKC);
                            if (AF(nplanes) > 2) store_value1<T>(PL2.dst, doff, vb, CKC);
                            if (AF(nplanes) > 3) store_value1<T>(PL3.dst, doff, vc, CKC);
#if GFW_CK
                            if (CK_INVC) ck_c64 += (unsigned long long)ck_bits1(va) + (AF(nplanes) > 2 ? ck_bits1(vb) : 0u) + (unsigned long long)(AF(nplanes) > 3 ? ck_bits1(vc) : 0u);
#endif
                            chroma_done = true;
                        }
                    }
                    if constexpr (FASTROW && I != 2 && !is_f32<T>::value && !INTERLEAVED_UV) if (fastrow && AF(nplanes) == 3) {
                        // bicubic / Lanczos4 chroma of planar frames: one vote, then both planes' interior taps with one set of bins
                        const int cbx = bin_x<I>(cu, PL1), cby = bin_y<I>(cv, PL1);
                        if (__builtin_expect(gfw_all_lanes(okm0 & lut_interior<T, I>(cbx, cby, PL1.w, PL1.h)), 1)) {
                            const uint32_t vu = inside_value1_lut<T, I>(PL1.src, PL1.src_stride, cbx, cby, &bg_c[0], lim_u, s_lut);
                            const uint32_t vv = inside_value1_lut<T, I>(PL2.src, PL1.src_stride, cbx, cby, &bg_v, lim_v, s_lut);
                            const int doff = row_off(cy, PL1.dst_stride) + cx * (int)sizeof(T);
                            store_value1<T>(PL1.dst, doff, vu, CKA); store_value1<T>(PL2.dst, doff, vv, CKA);
                            chroma_done = true;
                        }
                    }
                    if (fastrow && !chroma_done) ok0 = gfw_vote_lane(okm0, lane);       // the edge-aware samplers below take the per-lane form
                    if (chroma_done) {}
                    else if (I == 2) {
                        if (INTERLEAVED_UV) sample_store2<T, 2>(cu, cv, ok0, PL1, bg_c, lim_u, cx, cy, AUDIT ? AF(audit) : nullptr);
                        else if (AF(nplanes) == 3) sample_store_uv2<T>(cu, cv, ok0, PL1, PL2, bg_c[0], bg_v, lim_u, lim_v, cx, cy, AUDIT ? AF(audit) : nullptr);
                        else if (GFW_BAKE) sample_store_shared2_refs<T>(cu, cv, ok0, PL1, PL2, PL3, AF(nplanes) - 1, cx, cy);
                        else sample_store_shared2<T>(cu, cv, ok0, A_in.pl, 1, AF(nplanes) - 1, cx, cy);
                    } else {
                        if (INTERLEAVED_UV) sample_store<T, 2, I>(cu, cv, ok0, PL1, bg_c, lim_u, cx, cy, s_lut);
                        else if (GFW_BAKE) sample_store_shared_refs<T, I>(cu, cv, ok0, PL1, PL2, PL3, AF(nplanes) - 1, cx, cy, s_lut);
                        else sample_store_shared<T, I>(cu, cv, ok0, A_in.pl, 1, AF(nplanes) - 1, cx, cy, s_lut);
                    }
                }
#if GFW_TIMELINE >= 2
                if (fastrow) { GFW_TLB(3); tl_blk[4] += 1; }
#endif
            }
        }
        if (FAST1 && two_pass) __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");   // s_rows is rewritten by the next tile
#if GFW_PRIO_MODE
        --tiles_left;
#endif
#if GFW_TIMELINE
        { const unsigned long long tl_c = __builtin_readcyclecounter(); tl_p1 += tl_b - tl_a; tl_p3 += tl_c - tl_b; tl_units += (unsigned long long)RB; }
#endif
    }
#if GFW_CK
    ck_flush(cur_frame, n_frames);
#endif
#if GFW_TIMELINE
    if (lane == 0) {       // per wave: start, end (100 MHz device clock), phase clocks, lane-rows, HW_ID, XCC_ID, workgroup
        unsigned long long *o = gfw_tl + ((size_t)blockIdx.x * 4 + wave) * 8;
        o[0] = tl_start; o[1] = wall_clock64(); o[2] = tl_p1; o[3] = tl_p3; o[4] = tl_units;
        o[5] = __builtin_amdgcn_s_getreg(4 | (31 << 11)); o[6] = __builtin_amdgcn_s_getreg(20 | (31 << 11)); o[7] = blockIdx.x | ((tl_ready - tl_start) << 32);
#if GFW_TIMELINE >= 2
        unsigned long long *ob = gfw_tl_blocks + ((size_t)blockIdx.x * 4 + wave) * 8;
        for (int k = 0; k < 5; ++k) ob[k] = tl_blk[k];
        ob[5] = tl_p1; ob[6] = tl_p3; ob[7] = tl_units;
#endif
    }
#endif
}

#if GFW_JIT
}  // namespace
// The one instantiation a run-time build contains: template arguments and the bake header come from gfw_jit.hip.
extern "C" __global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(GFW_JIT_WAVES, 8))) void gfw_jit_kernel(const GfwClipArgs C) {
#ifndef GFW_JIT_AUDIT
#define GFW_JIT_AUDIT 0          // 1: the audit instantiation (GFW_OPT_KERNEL_VARIANT 3 / 4 on a clip whose certified first pass exists only in specialised builds)
#endif
    gfw_yuv_body<GFW_JIT_MODEL, GFW_JIT_T, GFW_JIT_N0, GFW_FRAME_TAPS, GFW_JIT_DW, GFW_JIT_DH, (GFW_JIT_IL != 0), GFW_JIT_RB, (GFW_JIT_FAST1 != 0), (GFW_JIT_AUDIT != 0)>(C.Y, &C);
}
#else
// Register budget: the specialised-fisheye instantiations are held to GFW_WAVES_PER_EU waves per SIMD; the generic-model ones (every
// other lens, digital lenses, refraction, IBIS/OIS, lens-correction blend) to GFW_GENERIC_WAVES_PER_EU (tools/kernel_resources.py
// lists what each instantiation takes; tests/test_kernel_resources.py pins it).
template <int MODEL, typename T, int N0, int I, int DW, int DH, bool INTERLEAVED_UV, int RB, bool FAST1, bool AUDIT>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(MODEL == GFW_MODEL_OPENCV_FISHEYE ? GFW_WAVES_PER_EU : GFW_GENERIC_WAVES_PER_EU, 8))) void gfw_yuv_kernel(const GfwYuvArgs A) {
    gfw_yuv_body<MODEL, T, N0, I, DW, DH, INTERLEAVED_UV, RB, FAST1, AUDIT>(A, nullptr);
}



template <int MODEL, typename T, int N0, int I, int RB, bool FAST1, bool AUDIT>
hipError_t launch_mt(const GfwYuvArgs &A, int dw, int dh, bool interleaved, hipStream_t s) {
    const int n_tiles = A.tiles_x * A.tiles_y;
    if (n_tiles <= 0) return hipSuccess;
    // persistent grid: a multiple of 8 (XCD bands), ~6 workgroups per CU by default, never more than one per tile
    int grid = A.grid_limit > 0 ? A.grid_limit : 256 * 6;
    const int per_xcd = (n_tiles + 7) >> 3;
    if (grid > per_xcd * 8) grid = per_xcd * 8;
    grid = (grid + 7) & ~7;
    dim3 block(64, 4);
#define GFW_YUV_LAUNCH(DW, DH, IL) hipLaunchKernelGGL((gfw_yuv_kernel<MODEL, T, N0, I, DW, DH, IL, RB, FAST1, AUDIT>), dim3(grid), block, 0, s, A)
#if GFW_HOT_ONLY
    if (dw == 2 && dh == 1 && !interleaved) GFW_YUV_LAUNCH(2, 1, false); else return hipErrorInvalidValue;
#else
    if constexpr (N0 > 1 || is_f32<T>::value) {          // packed single plane, or planar f32 planes: full resolution only
        if (dw == 1 && dh == 1 && !interleaved) GFW_YUV_LAUNCH(1, 1, false);
        else return hipErrorInvalidValue;
    } else {
        if (dw == 2 && dh == 1 && !interleaved) GFW_YUV_LAUNCH(2, 1, false);
        else if (dw == 2 && dh == 1 && interleaved) GFW_YUV_LAUNCH(2, 1, true);
        else if (dw == 2 && dh == 2 && !interleaved) GFW_YUV_LAUNCH(2, 2, false);
        else if (dw == 2 && dh == 2 && interleaved) GFW_YUV_LAUNCH(2, 2, true);
        else if (dw == 1 && dh == 1 && !interleaved) GFW_YUV_LAUNCH(1, 1, false);
        else if (dw == 1 && dh == 1 && interleaved) GFW_YUV_LAUNCH(1, 1, true);
        else return hipErrorInvalidValue;
    }
#endif
#undef GFW_YUV_LAUNCH
    return hipGetLastError();
}

}  // namespace

template <int MODEL, typename T, int N0>
static hipError_t launch_tn(const GfwYuvArgs &A, int dw, int dh, bool interleaved, bool fast1, hipStream_t s) {
    constexpr int I = GFW_FRAME_TAPS;
    if constexpr (MODEL == GFW_MODEL_OPENCV_FISHEYE) if (fast1) {      // the certified first pass exists for the specialised fisheye model only
        if (I == 2 && A.audit) return launch_mt<MODEL, T, N0, I, GFW_YUV_RB_FAST, true, (I == 2)>(A, dw, dh, interleaved, s);
        return launch_mt<MODEL, T, N0, I, GFW_YUV_RB_FAST, true, false>(A, dw, dh, interleaved, s);
    }
    return launch_mt<MODEL, T, N0, I, GFW_YUV_RB_EXACT, false, false>(A, dw, dh, interleaved, s);
}
// This translation unit is compiled once per (sample kind, tap count): -DGFW_FRAME_KIND=1|2|3|4 (u8, u16, f16, f32) and
// -DGFW_FRAME_TAPS=2|4|8 (bilinear, bicubic, Lanczos4), so that the nine families of instantiations build in parallel;
// gfw_kernels.hip dispatches on both.
#if !defined(GFW_FRAME_KIND) || !defined(GFW_FRAME_TAPS)
#error "compile with -DGFW_FRAME_KIND=1|2|3|4 -DGFW_FRAME_TAPS=2|4|8"
#endif
template <int MODEL>
static hipError_t launch_m(const GfwYuvArgs &A, int n0, int dw, int dh, bool interleaved, bool fast1, hipStream_t s) {
#if GFW_FRAME_KIND == 1
    if (n0 == 1) return launch_tn<MODEL, uint8_t, 1>(A, dw, dh, interleaved, fast1, s);
    if (n0 == 3) return launch_tn<MODEL, uint8_t, 3>(A, dw, dh, interleaved, fast1, s);
    if (n0 == 4) return launch_tn<MODEL, uint8_t, 4>(A, dw, dh, interleaved, fast1, s);
#elif GFW_FRAME_KIND == 2
    if (n0 == 1) return launch_tn<MODEL, uint16_t, 1>(A, dw, dh, interleaved, fast1, s);
    if (n0 == 3) return launch_tn<MODEL, uint16_t, 3>(A, dw, dh, interleaved, fast1, s);
    if (n0 == 4) return launch_tn<MODEL, uint16_t, 4>(A, dw, dh, interleaved, fast1, s);
#elif GFW_FRAME_KIND == 3
    if (n0 == 4) return launch_tn<MODEL, _Float16, 4>(A, dw, dh, interleaved, fast1, s);      // packed RGBAf16
#else
    if (n0 == 1) return launch_tn<MODEL, float, 1>(A, dw, dh, interleaved, fast1, s);
    if (n0 == 4) return launch_tn<MODEL, float, 4>(A, dw, dh, interleaved, fast1, s);
#endif
    return hipErrorInvalidValue;
}

#define GFW_CAT2(a, b) a##b
#define GFW_CAT(a, b) GFW_CAT2(a, b)
#define GFW_FN GFW_CAT(GFW_CAT(gfw_launch_yuv_kind, GFW_FRAME_KIND), GFW_CAT(_taps, GFW_FRAME_TAPS))
hipError_t GFW_FN(const GfwYuvArgs &A, int n0, int dw, int dh, bool interleaved, bool fast1, hipStream_t s) {
#if GFW_HOT_ONLY
    if (A.model == GFW_MODEL_OPENCV_FISHEYE && !A.extras && n0 == 1 && dw == 2 && dh == 1 && !interleaved && fast1 && !A.audit && GFW_FRAME_KIND == 2 && GFW_FRAME_TAPS == 2) {
        const hipError_t e = launch_mt<GFW_MODEL_OPENCV_FISHEYE, uint16_t, 1, 2, GFW_YUV_RB_FAST, true, false>(A, dw, dh, interleaved, s);
#if GFW_TIMELINE
        static int n_launch = 0;
        if (++n_launch == 60 && getenv("GFW_TIMELINE_FILE")) {
            static unsigned long long host[8192 * 8];
            (void)hipStreamSynchronize(s);
            (void)hipMemcpyFromSymbol(host, HIP_SYMBOL(gfw_tl), sizeof(host));
            if (FILE *f = fopen(getenv("GFW_TIMELINE_FILE"), "wb")) { fwrite(host, 1, sizeof(host), f); fclose(f); }
        }
#endif
        return e;
    }
    return hipErrorInvalidValue;
#else
    if (A.model == GFW_MODEL_OPENCV_FISHEYE && !A.extras) return launch_m<GFW_MODEL_OPENCV_FISHEYE>(A, n0, dw, dh, interleaved, fast1, s);
    if (A.extras & (16 | 32)) return launch_m<GFW_MODEL_GENERIC_EXTRA>(A, n0, dw, dh, interleaved, false, s);   // background mode 3 / Sony mesh: own instantiation
    return launch_m<-1>(A, n0, dw, dh, interleaved, false, s);
#endif
}
#endif   // !GFW_JIT
