#pragma once
// gfw_plane_kernel.h — the per-plane gfx950 kernel of libgfwarp (instantiated per PixelType in gfw_plane_inst.hip).
//
//   gfw_plane_kernel<PIX, I, MODEL>   one plane per launch; complete operator
//       (every PixelType, LUT + EWA samplers, all background modes, input rotation, digital lenses,
//       IBIS/mesh terms).  Behavioural replacement of the reference's OpenCL `undistort_image`
//       (src/core/gpu/opencl_undistort.cl:582-659) with the arithmetic of the CPU kernel
//       (src/core/stabilization/cpu_undistort.rs:233-633).
//
// Launch shape: 64x4 threads = one wave per output row segment of 64 pixels, 4 rows per workgroup, so a
// wave's stores are one contiguous 64*bpp-byte run and its gathers fall in a few adjacent source lines.
// blockIdx is swizzled so that consecutive workgroups of one XCD walk horizontally adjacent tiles
// (workgroup b lands on XCD b % 8: MI355X_MICROARCH.md), keeping a source-line neighbourhood in one L2.
#include <hip/hip_runtime.h>
#include "gfw_warp.h"
#include "gfw_launch.h"

// 32-phase tap table, one constant copy per translation unit (no relocatable device code needed)
namespace { __device__
#include "gfw_coeffs.inc"
}

// XCD-aware tile order: workgroup b runs on XCD (b % 8).  Give each XCD a contiguous band of tile rows.
__device__ __forceinline__ void gfw_tile_coords(int tiles_x, int tiles_y, int &tx, int &ty) {
    const int b = blockIdx.x;
    const int n = tiles_x * tiles_y;
    const int xcd = b & 7, j = b >> 3;
    const int per = (n + 7) >> 3;                    // tiles per XCD band
    int t = xcd * per + j;
    if (t >= n) { tx = -1; ty = -1; return; }
    ty = t / tiles_x; tx = t - ty * tiles_x;
}

// DUAL (EWA only, round 6): the launch also warps a second plane that shares every kernel parameter but the background — U and V of a planar frame: their
// coordinates, jacobians and tap weights are the same numbers, so they are worked out once and only the sums are kept twice (cpu_undistort.rs:331-369 per plane:
// the same operations on the same operands in the same order for each of the two).  The host decides (gfw_api.hip run_planes).
template <int PIX, int I, int MODEL, bool DUAL = false>
__global__ __launch_bounds__(256) void gfw_plane_kernel(const GfwPlane A, const GfwCommon C) {
    constexpr int N = GfwPix<PIX>::N;
    constexpr int BPP = GfwPix<PIX>::BPP;
    __shared__ float lut[I == 4 || I == 8 ? 448 : 1];
    if (I == 4 || I == 8) {
        for (int i = threadIdx.y * 64 + threadIdx.x; i < 448; i += 256) lut[i] = GFW_COEFFS[i];
        __syncthreads();
    }
    const gfw_kernel_params &P = A.p;
    const int tiles_x = (A.out_cols + 63) >> 6, tiles_y = (A.out_rows + 3) >> 2;
    int tx, ty;
    gfw_tile_coords(tiles_x, tiles_y, tx, ty);
    if (tx < 0) return;
    const int x = tx * 64 + threadIdx.x, y = ty * 4 + threadIdx.y;
    if (x >= A.out_cols || y >= A.out_rows) return;
    if ((int64_t)y * A.dst_stride + (int64_t)(x + 1) * BPP > A.dst_len) return;   // partial trailing row chunk

    // :546-551 — position in output space; pixels outside are never written
    const float opx = gfw_map_coord((float)x, (float)P.output_rect[0], (float)(P.output_rect[0] + P.output_rect[2]), 0.0f, (float)P.output_width);
    const float opy = gfw_map_coord((float)y, (float)P.output_rect[1], (float)(P.output_rect[1] + P.output_rect[3]), 0.0f, (float)P.output_height);
    if (!(opx >= 0.0f && opy >= 0.0f && gfw_f2i(opx) < P.output_width && gfw_f2i(opy) < P.output_height)) return;

    uint8_t *pix_out = A.dst + (int64_t)y * A.dst_stride + (int64_t)x * BPP;
    uint8_t *pix_out2 = DUAL ? A.dst2 + (int64_t)y * A.dst_stride + (int64_t)x * BPP : nullptr;
    float bg[N], pixel[N], bg2[N], pixel2[N];
    #pragma unroll
    for (int c = 0; c < N; ++c) { bg[c] = P.background[c] * P.max_pixel_value; pixel[c] = bg[c]; bg2[c] = DUAL ? A.background2[c] * P.max_pixel_value : 0.0f; pixel2[c] = bg2[c]; }     // :523
    if ((P.flags & 4) == 4) { GfwPix<PIX>::store(pix_out, bg); if constexpr (DUAL) GfwPix<PIX>::store(pix_out2, bg2); return; }                              // :558-561
    const bool fix_range = (P.flags & 1) == 1, is_y = P.plane_index == 0;

    GfwPt uv = gfw_undistort_coord_fullres<MODEL>((float)x, (float)y, P, C);
    if (uv.ok) {
        float jac[4] = {1.0f, 0.0f, 0.0f, 1.0f};
        if (I == 0) {                                                                                  // :567-572
            const float eps = 0.01f;
            GfwPt a = gfw_undistort_coord_fullres<MODEL>((float)x + eps, (float)y, P, C);
            GfwPt b = gfw_undistort_coord_fullres<MODEL>((float)x, (float)y + eps, P, C);
            float ux = uv.x, uy = uv.y;
            if (P.background_mode != 3) {
                gfw_to_source_rect(ux, uy, P, C);
                if (a.ok) gfw_to_source_rect(a.x, a.y, P, C);
                if (b.ok) gfw_to_source_rect(b.x, b.y, P, C);
            }
            const float ax = a.ok ? a.x : 0.0f, ay = a.ok ? a.y : 0.0f, bx = b.ok ? b.x : 0.0f, by = b.ok ? b.y : 0.0f;
            jac[0] = (ax - ux) / eps; jac[1] = (bx - ux) / eps; jac[2] = (ay - uy) / eps; jac[3] = (by - uy) / eps;
        }
        if (P.background_mode == 3) {                                                                  // :576-613
            const float width_f = (float)P.width, height_f = (float)P.height;
            const float widthf = width_f - 1.0f, heightf = height_f - 1.0f;
            const float feather = gfw_max(P.background_margin_feather * heightf, 0.0001f);
            float p2x = uv.x, p2y = uv.y, alpha = 1.0f;
            if ((uv.x > widthf - feather) || (uv.x < feather) || (uv.y > heightf - feather) || (uv.y < feather)) {
                alpha = gfw_max(gfw_min(gfw_min(gfw_min(gfw_min(widthf - uv.x, heightf - uv.y), uv.x), uv.y) / feather, 1.0f), 0.0f);
                p2x = p2x / width_f; p2y = p2y / height_f;
                p2x = ((p2x - 0.5f) * (1.0f - P.background_margin)) + 0.5f;
                p2y = ((p2y - 0.5f) * (1.0f - P.background_margin)) + 0.5f;
                p2x = p2x * width_f; p2y = p2y * height_f;
            }
            float ux = uv.x, uy = uv.y;
            gfw_to_source_rect(ux, uy, P, C);
            gfw_to_source_rect(p2x, p2y, P, C);
            float c1[N], c2[N], d1[N], d2[N];
            gfw_sample<PIX, I, DUAL>(ux, uy, jac, P, A.src, bg, lut, c1, A.src2, bg2, d1);
            gfw_sample<PIX, I, DUAL>(p2x, p2y, jac, P, A.src, bg, lut, c2, A.src2, bg2, d2);
            #pragma unroll
            for (int c = 0; c < N; ++c) { pixel[c] = c1[c] * alpha + c2[c] * (1.0f - alpha); if constexpr (DUAL) pixel2[c] = d1[c] * alpha + d2[c] * (1.0f - alpha); }
            if (fix_range) { gfw_remap_colorrange<N>(pixel, is_y); if constexpr (DUAL) gfw_remap_colorrange<N>(pixel2, is_y); }
            GfwPix<PIX>::store(pix_out, pixel);
            if constexpr (DUAL) GfwPix<PIX>::store(pix_out2, pixel2);
            return;
        }
        gfw_to_source_rect(uv.x, uv.y, P, C);                                                          // :510-515
        gfw_sample<PIX, I, DUAL>(uv.x, uv.y, jac, P, A.src, bg, lut, pixel, A.src2, bg2, pixel2);
    }
    if (fix_range) { gfw_remap_colorrange<N>(pixel, is_y); if constexpr (DUAL) gfw_remap_colorrange<N>(pixel2, is_y); }
    GfwPix<PIX>::store(pix_out, pixel);
    if constexpr (DUAL) GfwPix<PIX>::store(pix_out2, pixel2);
}


// ---------------------------------------------------------------------------- launchers
template <int PIX, int I>
static hipError_t launch_plane_pi(const GfwPlane &A, const GfwCommon &C, hipStream_t s) {
    const int tiles_x = (A.out_cols + 63) >> 6, tiles_y = (A.out_rows + 3) >> 2;
    const int n = tiles_x * tiles_y;
    const int grid = ((n + 7) >> 3) << 3;
    if (grid <= 0) return hipSuccess;
    dim3 block(64, 4);
    if constexpr (I == 0 && GfwPix<PIX>::N == 1) {
        if (A.src2) {                                      // EWA on a pair of single-channel planes (the host paired them: run_planes)
            if (C.model == GFW_MODEL_OPENCV_FISHEYE && C.mesh_len == 0)
                hipLaunchKernelGGL((gfw_plane_kernel<PIX, I, GFW_MODEL_OPENCV_FISHEYE, true>), dim3(grid), block, 0, s, A, C);
            else
                hipLaunchKernelGGL((gfw_plane_kernel<PIX, I, -1, true>), dim3(grid), block, 0, s, A, C);
            return hipGetLastError();
        }
    }
    if (A.src2) return hipErrorInvalidValue;               // (never paired by the host for anything else)
    if (C.model == GFW_MODEL_OPENCV_FISHEYE && C.mesh_len == 0)
        hipLaunchKernelGGL((gfw_plane_kernel<PIX, I, GFW_MODEL_OPENCV_FISHEYE>), dim3(grid), block, 0, s, A, C);
    else
        hipLaunchKernelGGL((gfw_plane_kernel<PIX, I, -1>), dim3(grid), block, 0, s, A, C);
    return hipGetLastError();
}
template <int PIX>
static hipError_t launch_plane_p(const GfwPlane &A, const GfwCommon &C, hipStream_t s) {
    switch (A.p.interpolation) {
    case 2: return launch_plane_pi<PIX, 2>(A, C, s);
    case 4: return launch_plane_pi<PIX, 4>(A, C, s);
    case 8: return launch_plane_pi<PIX, 8>(A, C, s);
    case 10: case 11: case 12: case 13: return launch_plane_pi<PIX, 0>(A, C, s);
    default: return hipErrorInvalidValue;
    }
}
