// gfw_plane_inst.hip — instantiates gfw_plane_kernel for ONE PixelType (-DGFW_PIX_ID=n); the build
// compiles this file once per pixel type in parallel (13 objects) instead of one 4-minute translation unit.
#include "gfw_plane_kernel.h"
#ifndef GFW_PIX_ID
#error "compile with -DGFW_PIX_ID=<GFW_PIX_* value>"
#endif
#define GFW_CAT2(a, b) a##b
#define GFW_CAT(a, b) GFW_CAT2(a, b)
hipError_t GFW_CAT(gfw_launch_plane_pix, GFW_PIX_ID)(const GfwPlane &A, const GfwCommon &C, hipStream_t s) {
    return launch_plane_p<GFW_PIX_ID>(A, C, s);
}
