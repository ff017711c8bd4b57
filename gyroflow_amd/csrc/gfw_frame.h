// gfw_frame.h — argument block + launcher of the fused YUV frame kernel (gfw_frame.hip)
#pragma once
#if !defined(GFW_JIT) || !GFW_JIT
#include <hip/hip_runtime.h>
#endif
#include "gfw_warp.h"
#include "gfw_fastmath.h"

struct GfwYuvPlane {
    const uint8_t *src;
    uint8_t *dst;
    int32_t src_stride, dst_stride;   // bytes
    int32_t w, h;                     // source plane size (= source_rect w,h)
    float bg[4];                      // background[c] * max_pixel_value
    float limit;                      // pixel_value_limit
    int32_t src_len, dst_len;         // bytes the caller declared for the two buffers (< 2 GiB on this path): audit mode range-checks against them
    int32_t fix;                      // FIX_COLOR_RANGE (flags & 1, cpu_undistort.rs:254-260, :619-621): 0 off, 1 the luma scale (plane_index 0), 2 the chroma scale
    float org_x, org_y;               // source_rect origin (mod.rs:322: a BufferDescription's rect): added to the mapped coordinate as map_coord adds it (cpu_undistort.rs:510-515) ...
    int32_t ox32, oy32;               // ... and 32 * origin, which comes off the 1/32-pixel bins again: src / dst point at the rects' first pixels, w / h are the rects'
};

#define GFW_P1_TABLE_N 8192      // intervals of the s(rho) table of the certified first pass (64 KB)
#define GFW_YUV_RB_FAST 4         // luma block rows per lane with the certified first pass (tile = 64 x 16 lanes-rows)
#define GFW_YUV_RB_EXACT 1

struct GfwYuvArgs {
    GfwYuvPlane pl[4];
    const float *matrices;            // [matrix_count][GFW_MAT_STRIDE]
    int32_t nplanes;
    int32_t width, height;            // KernelParams.width/height (full-res source)
    int32_t out_w, out_h;             // luma output plane size (= output_width/height)
    int32_t cw, ch;                   // thread grid: chroma-site counts (out_w/DW, out_h/DH, rounded up)
    int32_t tiles_x, tiles_y;
    int32_t matrix_count;
    int32_t hrs;                      // flags & 16
    int32_t model;
    int32_t k_all_zero;               // k[0..3] all zero (opencv_fisheye.rs:75)
    int32_t hstretch_div, vstretch_div;
    int32_t fix_range;                // some plane has `fix` set: the frame takes the per-pixel path (the range fix sits between the sample and the store)
    int32_t checksum;                 // specialised builds only (GFW_BK_checksum): the kernel takes the checksum of what it writes (gfw_set_frame_checksums) into ck_part
    int32_t fill_bg;                  // FILL_WITH_BACKGROUND (flags & 4, cpu_undistort.rs:558-561): every pixel of the output rect is the background
    int32_t rot_on;                   // input_rotation != 0 (:485-491): the projected point is rotated about the frame centre (cos / sin / rotated frame size in `common`)
    int32_t background_mode;          // 0 solid, 1 edge repeat, 2 edge mirror, 3 margin + feather (with extras & 16)
    int32_t grid_limit;               // persistent workgroups to launch (0 = default)
    int32_t extras;                   // features served by the generic-model instantiation only: 1 IBIS/OIS terms in the
                                      // matrix rows, 2 digital lens, 4 light refraction (cpu_undistort.rs:143-165, :216-220),
                                      // 8 lens-correction blend (lens_correction_amount < 1, :429-460),
                                      // 16 background mode 3: margin with feather (:576-613),
                                      // 32 Sony lens-distortion mesh / focal-plane distortion in `common.mesh` (:169-214)
    int32_t ablate;                   // always 0 from the library; read only by GFW_TESTING builds of the kernel (the timing ablations of tools/ A/B runs: gfw_frame.hip)
    float hstretch, vstretch;
    float f[2], c[2], k[12];
    float t2[2];
    float r_limit_sq;
    GfwMapConst map_lx, map_ly, map_cx, map_cy;
    // certified first pass (gfw_frame.hip): table of (s_i, s_{i+1}-s_i) over rho in [0, rho_max], certificate half-width
    const float2 *p1_table;
    float p1_rho_max, p1_rho_scale;   // scale = N / rho_max (rho form), N / r_max (r form: p1_rform)
    float p1_kmax; int32_t p1_rform;  // the table's last key: rho_max, or r_max = sqrt(rho_max) when the table runs over r (radial models other than the fisheye; gfw_frame.hip p1_key)
    float p1_eps, p1_ew, p1_em;       // E = p1_eps + p1_ew * omega + p1_em * mu: bound on |approx - exact| of the projected row/column coordinate, pixels;
                                      // omega, mu = the cancellation measures of the frame's mid-row matrix, evaluated by the kernel
    float p1_f, p1_c;                 // f[1], c[1] (f[0], c[0] for horizontal rolling shutter)
    float p1_lat[6];                  // lattice form of the first pass: bounds max |s|, max sqrt(rho) |s'|, max rho |s'|, max rho^1.5 |s''| over the table's range
                                      // (gfw_api.hip p1_prepare_table), the interpolation's own rounding allowance in pixels, [5] != 0: per-pixel form on request
    unsigned long long *ck_part;      // checksum builds: partial sums of the launch, [frame][workgroup][wave] (one word each; gfw_ck_finish adds a frame's words to its sum)
    unsigned long long *audit;        // nullptr, or 8 words: certified, certified-but-wrong, queued, queue-overflow, max |approx-exact| (f32 bits),
                                      // [5] global addresses outside their buffer (audit mode range-checks every tap, store, matrix row and table entry)
    gfw_kernel_params kp;             // plane-0 params, for the non-specialised lens models
    GfwCommon common;
};

// Several frames of one clip in one launch (run-time-specialised kernel only): everything but these pointers is shared.
#define GFW_CLIP_MAX 16         // frames per launch (= GFW_CLIP_FRAMES_MAX of gfwarp.h)
struct GfwFrameDyn { const uint8_t *src[4]; uint8_t *dst[4]; const float *matrices; };
struct GfwClipArgs { GfwYuvArgs Y; int32_t n_frames; int32_t pad_; GfwFrameDyn fr[GFW_CLIP_MAX]; };

#if !defined(GFW_JIT) || !GFW_JIT
int gfw_yuv_rows_per_lane(bool fast1, int tune_rb);
hipError_t gfw_launch_yuv(const GfwYuvArgs &A, int sample_kind, int taps, int n0, int dw, int dh, bool interleaved, bool fast1, hipStream_t s);
#endif
