// gfw_launch.h — host-visible launch entry points of gfw_kernels.hip
#pragma once
#include <hip/hip_runtime.h>
#include "gfw_warp.h"

hipError_t gfw_launch_plane(const GfwPlane &A, const GfwCommon &C, hipStream_t s);
hipError_t gfw_launch_repack(const float *in, float *out, int rows, hipStream_t s);
hipError_t gfw_launch_debug_math(int op, const float *a, const float *b, float *out, size_t n, hipStream_t s);
hipError_t gfw_launch_debug_selftest(int test, unsigned long long n, unsigned long long seed, unsigned long long *bad, hipStream_t s);
hipError_t gfw_launch_stmap(const gfw_kernel_params &P, const GfwCommon &C, int width, int height, float *coords, hipStream_t s);
