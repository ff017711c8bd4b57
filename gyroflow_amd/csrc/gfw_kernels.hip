// gfw_kernels.hip — kernel dispatch by PixelType + the small utility kernels of libgfwarp.
#include <hip/hip_runtime.h>
#include "gfw_launch.h"

#define GFW_DECL(n) hipError_t gfw_launch_plane_pix##n(const GfwPlane &A, const GfwCommon &C, hipStream_t s);
GFW_DECL(0) GFW_DECL(1) GFW_DECL(2) GFW_DECL(3) GFW_DECL(4) GFW_DECL(5) GFW_DECL(6)
GFW_DECL(7) GFW_DECL(8) GFW_DECL(9) GFW_DECL(10) GFW_DECL(11) GFW_DECL(12)

hipError_t gfw_launch_plane(const GfwPlane &A, const GfwCommon &C, hipStream_t s) {
    switch (A.pix) {
#define GFW_CASE(n) case n: return gfw_launch_plane_pix##n(A, C, s);
    GFW_CASE(0) GFW_CASE(1) GFW_CASE(2) GFW_CASE(3) GFW_CASE(4) GFW_CASE(5) GFW_CASE(6)
    GFW_CASE(7) GFW_CASE(8) GFW_CASE(9) GFW_CASE(10) GFW_CASE(11) GFW_CASE(12)
    default: return hipErrorInvalidValue;
    }
}

// Row repack: [rows][14] f32 (FrameTransform.matrices) -> [rows][16] with cos(-m11), sin(-m11) slots.
// Used only for device-resident matrices; the trig slots are 1 / 0 (no IBIS roll), see gfw_api.hip.
__global__ void gfw_repack_matrices_kernel(const float *in, float *out, int rows) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= rows * GFW_MAT_STRIDE) return;
    const int r = i >> 4, c = i & 15;
    out[i] = c < 14 ? in[r * 14 + c] : (c == 14 ? 1.0f : 0.0f);
}
hipError_t gfw_launch_repack(const float *in, float *out, int rows, hipStream_t s) {
    const int n = rows * GFW_MAT_STRIDE;
    hipLaunchKernelGGL(gfw_repack_matrices_kernel, dim3((n + 255) / 256), dim3(256), 0, s, in, out, rows);
    return hipGetLastError();
}
