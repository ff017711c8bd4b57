// Stub launchers for the lean A/B libraries of tools/build_variants.sh: every launcher the variant's own translation unit does not
// define answers hipErrorInvalidValue (weak symbols: the variant's strong definition wins), so that a variant library is a few
// hundred KB instead of 33 MB (benchmarking only, never shipped).
#include <hip/hip_runtime.h>
#include "../gyroflow_amd/csrc/gfw_launch.h"
#include "../gyroflow_amd/csrc/gfw_frame.h"
#define GFW_STUB_PLANE(n) __attribute__((weak)) hipError_t gfw_launch_plane_pix##n(const GfwPlane &, const GfwCommon &, hipStream_t) { return hipErrorInvalidValue; }
GFW_STUB_PLANE(0) GFW_STUB_PLANE(1) GFW_STUB_PLANE(2) GFW_STUB_PLANE(3) GFW_STUB_PLANE(4) GFW_STUB_PLANE(5) GFW_STUB_PLANE(6)
GFW_STUB_PLANE(7) GFW_STUB_PLANE(8) GFW_STUB_PLANE(9) GFW_STUB_PLANE(10) GFW_STUB_PLANE(11) GFW_STUB_PLANE(12)
#define GFW_STUB_YUV(K, I) __attribute__((weak)) hipError_t gfw_launch_yuv_kind##K##_taps##I(const GfwYuvArgs &, int, int, int, bool, bool, hipStream_t) { return hipErrorInvalidValue; }
GFW_STUB_YUV(1, 2) GFW_STUB_YUV(1, 4) GFW_STUB_YUV(1, 8) GFW_STUB_YUV(2, 2) GFW_STUB_YUV(2, 4) GFW_STUB_YUV(2, 8) GFW_STUB_YUV(4, 2) GFW_STUB_YUV(4, 4) GFW_STUB_YUV(4, 8)
