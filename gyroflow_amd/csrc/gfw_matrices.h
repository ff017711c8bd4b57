// gfw_matrices.h — device-side per-row matrix builder (gfw_matrices.hip)
#pragma once
#include <hip/hip_runtime.h>
#include "../../include/gfwarp.h"

struct GfwTracks {            // device-resident quaternion tracks (timestamp_us ascending; quats as w,x,y,z f64)
    const int64_t *org_ts; const double *org_q; int org_n;
    const int64_t *sm_ts;  const double *sm_q;  int sm_n;
};
hipError_t gfw_launch_build_matrices(const GfwTracks &T, const gfw_frame_timing &F, float *out, hipStream_t s);
