// gfw_frame.h — argument block + launcher of the fused YUV frame kernel (gfw_frame.hip)
#pragma once
#include <hip/hip_runtime.h>
#include "gfw_warp.h"
#include "gfw_fastmath.h"

struct GfwYuvPlane {
    const uint8_t *src;
    uint8_t *dst;
    int32_t src_stride, dst_stride;   // bytes
    int32_t w, h;                     // source plane size (= source_rect w,h)
    float bg[2];                      // background[c] * max_pixel_value
    float limit;                      // pixel_value_limit
    int32_t pad_;
};

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
    float hstretch, vstretch;
    float f[2], c[2], k[12];
    float t2[2];
    float r_limit_sq;
    GfwMapConst map_lx, map_ly, map_cx, map_cy;
    gfw_kernel_params kp;             // plane-0 params, for the non-specialised lens models
    GfwCommon common;
};

hipError_t gfw_launch_yuv(const GfwYuvArgs &A, int bytes_per_sample, int dw, int dh, bool interleaved, hipStream_t s);
