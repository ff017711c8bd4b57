// gfw_launch.h — host-visible launch entry points of gfw_kernels.hip
#pragma once
#include <hip/hip_runtime.h>
#include "gfw_warp.h"

hipError_t gfw_launch_plane(const GfwPlane &A, const GfwCommon &C, hipStream_t s);
hipError_t gfw_launch_repack(const float *in, float *out, int rows, hipStream_t s);
hipError_t gfw_launch_checksum64(const void *buf, size_t bytes, unsigned long long *out, hipStream_t s);
// gfw_set_frame_checksums: the sums of a launch of the checksum build of the fused kernel (its table of partial sums, [frame][per_frame words]), and the pass over a
// written region (rows of row_bytes bytes from dst + first_byte, `stride` apart) that follows every other kernel
struct GfwCkSums { unsigned long long *sum[16]; };
hipError_t gfw_launch_ck_finish(const unsigned long long *part, int per_frame, int n_frames, const GfwCkSums &sums, hipStream_t s);
hipError_t gfw_launch_ck_region(const uint8_t *dst, long long first_byte, long long stride, int row_bytes, int rows, unsigned long long *out, hipStream_t s);
hipError_t gfw_launch_debug_math(int op, const float *a, const float *b, float *out, size_t n, hipStream_t s);
hipError_t gfw_launch_debug_selftest(int test, unsigned long long n, unsigned long long seed, unsigned long long *bad, hipStream_t s);
hipError_t gfw_launch_stmap(const gfw_kernel_params &P, const GfwCommon &C, int width, int height, float *coords, hipStream_t s);

struct GfwPointsArgs {
    const float *points;      // n x 2 f32 (device) or nullptr = pixel grid
    size_t n;
    int32_t grid_w;
    int32_t rotation_count;
    const float *rotations;   // [rotation_count][9] row-major f32 (device)
    const float *shifts;      // [rotation_count][6] or nullptr (device)
    int32_t index_mode;       // 0 single, 1 per point, 2 per grid row, 3 per grid column
    int32_t mesh_len;
    const double *mesh;       // f64 mesh (device) or nullptr
    float *out;               // n x 2 f32 (device)
};
hipError_t gfw_launch_points(const gfw_kernel_params &P, const GfwCommon &C, const GfwPointsArgs &A, hipStream_t s);
