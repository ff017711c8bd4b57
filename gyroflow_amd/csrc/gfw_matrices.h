// gfw_matrices.h — device-side per-row matrix builder (gfw_matrices.hip)
#pragma once
#include <hip/hip_runtime.h>
#include "../../include/gfwarp.h"

struct GfwTracks {            // device-resident quaternion tracks (timestamp_us ascending; quats as w,x,y,z f64)
    const int64_t *org_ts; const double *org_q; int org_n;
    const int64_t *sm_ts;  const double *sm_q;  int sm_n;
    // gyro/video sync (GyroSource.offsets_adjusted, duration_ms): gyro_source/mod.rs:857-860
    const int64_t *off_ts; const double *off_ms; int off_n;
    double duration_ms;
};
struct GfwStab {              // gfw_frame_stab with device-resident control points (x4 doubles each), or counts of -1: no data
    double offset, sensor_h, crop_y, crop_h, scale_x, scale_y;
    double height;
    const double *ibis, *ois;
    int ibis_n, ois_n;
};
// Builds `frames` tables of packed rows: d_timings[frames] (device), table f at out + f * table_floats, rows of frame f =
// d_timings[f].rows (<= max_rows).  prefix_scratch: 4 * frames doubles of device memory owned by the caller for the
// duration of the launch.
hipError_t gfw_launch_build_matrices(const GfwTracks &T, const gfw_frame_timing *d_timings, int frames, int max_rows, double *prefix_scratch,
                                     float *out, size_t table_floats, hipStream_t s, const GfwStab *stab = nullptr);
