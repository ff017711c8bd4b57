// gfw_matrices.hip — per-row rolling-shutter matrices built on the device.
//
// The step immediately before the warp kernel: FrameTransform::at_timestamp (src/core/stabilization/
// frame_transform.rs:221-308) evaluates, for every sensor row y,
//     quat  = smoothed(ts) * org(ts)^-1 * org(start_ts + row_readout_time * y)
//     R     = image_rotation * R(quat), with the framebuffer sign flips (:261-267)
//     row_y = f32( inv(new_k * R) )
// on the host with rayon + an f64 SVD pseudo-inverse and uploads 14 floats per row every frame.  Here one lane
// does one row in f64 — quaternion lookup with the reference's rounding/clamping and nalgebra's slerp
// (src/core/gyro_source/mod.rs:857-882), closed-form 3x3 inverse — and writes libgfwarp's packed 64-byte row straight
// into HBM, so the per-frame host->device traffic of this path drops from 121 KB to a 160-byte descriptor.
// Parity is tolerance-based by construction (SVD vs adjugate inverse, ocml vs libm acos/sin): tests compare the rows
// with the float64 host statement to <= 2 ULP of f32 and then warp with the device-built rows bit-exactly.
#include <hip/hip_runtime.h>
#include "gfw_warp.h"
#include "gfw_matrices.h"
#include "gfw_math.h"

namespace {

struct Q { double w, x, y, z; };
__device__ __forceinline__ Q qmul(const Q &a, const Q &b) {
    return Q{a.w * b.w - a.x * b.x - a.y * b.y - a.z * b.z, a.w * b.x + a.x * b.w + a.y * b.z - a.z * b.y,
             a.w * b.y - a.x * b.z + a.y * b.w + a.z * b.x, a.w * b.z + a.x * b.y - a.y * b.x + a.z * b.w};
}
// nalgebra UnitQuaternion::slerp (Unit<Vector4>::try_slerp with the shorter-arc flip)
__device__ __forceinline__ Q slerp(const Q &a, Q b, double t) {
    double c = a.w * b.w + a.x * b.x + a.y * b.y + a.z * b.z;
    if (c < 0.0) { b = Q{-b.w, -b.x, -b.y, -b.z}; c = -c; }
    if (fabs(c) >= 1.0) return a;
    const double hang = acos(c);
    const double s = sqrt(1.0 - c * c);
    if (s == 0.0) return a;
    const double ta = sin((1.0 - t) * hang) / s, tb = sin(t * hang) / s;
    return Q{a.w * ta + b.w * tb, a.x * ta + b.x * tb, a.y * ta + b.y * tb, a.z * ta + b.z * tb};
}
// Rust `f64 as i64`: truncate toward zero, saturate, NaN -> 0
__device__ __forceinline__ int64_t f2i64(double v) {
    if (!(v == v)) return 0;
    if (v >= 9223372036854775807.0) return INT64_MAX;
    if (v <= -9223372036854775808.0) return INT64_MIN;
    return (int64_t)v;
}
// GyroSource::offset_at_timestamp (gyro_source/mod.rs:884-908): linear interpolation (and extrapolation) of the sync offsets
__device__ double offset_at(const int64_t *ts, const double *v, int n, double timestamp_ms) {
    if (n <= 0) return 0.0;
    if (n == 1) return v[0];
    const int64_t timestamp_us = f2i64(timestamp_ms * 1000.0);
    int64_t lookup = timestamp_us;
    if (lookup > ts[n - 1] - 1) lookup = ts[n - 1] - 1;           // .min(last_ts - 1)
    if (lookup < ts[0] + 1) lookup = ts[0] + 1;                   // .max(first_ts + 1)
    if (lookup < ts[0]) return 0.0;                               // range(..=lookup) empty
    int lo = 0, hi = n - 1;                                       // last index with ts[i] <= lookup
    while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (ts[mid] <= lookup) lo = mid; else hi = mid - 1; }
    if (ts[lo] == lookup) return v[lo];
    if (lo + 1 >= n) return 0.0;                                  // range(lookup..) empty
    const double time_delta = (double)(ts[lo + 1] - ts[lo]);
    const double fract = (double)(timestamp_us - ts[lo]) / time_delta;
    return v[lo] + (v[lo + 1] - v[lo]) * fract;
}
// GyroSource::quat_at_timestamp (gyro_source/mod.rs:857-882) over a sorted (timestamp_us -> quaternion) track
__device__ Q quat_at(const GfwTracks &T, const int64_t *ts, const double *q, int n, double timestamp_ms) {
    if (n < 2 || !(T.duration_ms > 0.0)) return Q{1.0, 0.0, 0.0, 0.0};
    timestamp_ms -= offset_at(T.off_ts, T.off_ms, T.off_n, timestamp_ms);
    int64_t lookup = f2i64(round(timestamp_ms * 1000.0));
    if (lookup > ts[n - 1]) lookup = ts[n - 1];
    if (lookup < ts[0]) lookup = ts[0];
    int lo = 0, hi = n - 1;                     // last index with ts[i] <= lookup
    while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (ts[mid] <= lookup) lo = mid; else hi = mid - 1; }
    const Q q1{q[lo * 4], q[lo * 4 + 1], q[lo * 4 + 2], q[lo * 4 + 3]};
    if (ts[lo] == lookup || lo + 1 >= n) return q1;
    const Q q2{q[lo * 4 + 4], q[lo * 4 + 5], q[lo * 4 + 6], q[lo * 4 + 7]};
    const double fract = (double)(lookup - ts[lo]) / (double)(ts[lo + 1] - ts[lo]);
    return slerp(q1, q2, fract);
}

// The row-independent factor smoothed(ts) * org(ts)^-1 (frame_transform.rs:255-256,289-291), once per frame: a
// one-lane kernel in front of the row kernel, so that the 34 row waves do one slerp each instead of three.
// Batched form: `frames` frame descriptors in device memory, one prefix lane per frame, blockIdx.y = frame.
__global__ void gfw_build_prefix_kernel(const GfwTracks T, const gfw_frame_timing *Fs, int frames, double *prefix) {
    const int f = blockIdx.x * blockDim.x + threadIdx.x;
    if (f >= frames) return;
    const gfw_frame_timing &F = Fs[f];
    prefix += (size_t)f * 4;
    const double ts = F.timestamp_ms + F.per_frame_time_offset_ms;
    Q q1 = quat_at(T, T.org_ts, T.org_q, T.org_n, ts);
    const double n1 = q1.w * q1.w + q1.x * q1.x + q1.y * q1.y + q1.z * q1.z;
    q1 = Q{q1.w / n1, -q1.x / n1, -q1.y / n1, -q1.z / n1};                         // inverse()
    const Q sm = quat_at(T, T.sm_ts, T.sm_q, T.sm_n, ts);
    const Q pre = qmul(sm, q1);
    prefix[0] = pre.w; prefix[1] = pre.x; prefix[2] = pre.y; prefix[3] = pre.z;
}
// CatmullRom<Vector3<f64>>::interpolate (gyro_source/splines.rs:22-84) over `n` control points (position, x, y, z);
// false = None (the caller substitutes the default, zero)
__device__ bool catmull_rom_at(const double *pts, int n, double t, double out[3]) {
    if (n < 2 || !(t == t)) return false;
    // search_lower_cp: binary_search_by(partial_cmp): Ok(i) exact hit, Err(i) insertion point
    int lo = 0, hi = n;                                           // first index with position >= t
    while (lo < hi) { const int mid = (lo + hi) >> 1; if (pts[mid * 4] < t) lo = mid + 1; else hi = mid; }
    int lower;
    if (lo < n && pts[lo * 4] == t) { if (lo == n - 1) return false; lower = lo; }
    else { if (lo >= n || lo == 0) return false; lower = lo - 1; }
    if (lower + 1 >= n) return false;
    const double *pa = pts + (size_t)lower * 4, *pb = pa + 4;
    const double k = (t - pa[0]) / (pb[0] - pa[0]);               // normalize
    for (int c = 0; c < 3; ++c) {
        const double a = pa[1 + c], b = pb[1 + c];
        const double x = (lower <= 0) ? a * 2.0 - b : pa[1 + c - 4];
        const double y = (lower + 2 >= n) ? b * 2.0 - a : pb[1 + c + 4];
        out[c] = ((((a * 3.0 - x) - b * 3.0) + y) * 0.5) * k * k * k + ((b - x) * 0.5) * k + a + (((b * 4.0 + a * -5.0 + x + x) - y) * 0.5) * k * k;
    }
    return true;
}
__global__ void gfw_build_matrices_kernel(const GfwTracks T, const gfw_frame_timing *Fs, const double *prefix, float *out, size_t table_floats, const GfwStab S) {
    const gfw_frame_timing &F = Fs[blockIdx.y];
    prefix += (size_t)blockIdx.y * 4;
    out += (size_t)blockIdx.y * table_floats;
    const int y = blockIdx.x * blockDim.x + threadIdx.x;
    if (y >= F.rows) return;
    const double frt = F.frame_readout_time_ms;
    const double ts = F.timestamp_ms + F.per_frame_time_offset_ms;
    const double start_ts = ts - frt / 2.0;
    const double row_t = frt / (double)F.readout_dim;
    const double qt = (fabs(frt) > 0.0) ? start_ts + row_t * (double)y : start_ts;
    const Q pre{prefix[0], prefix[1], prefix[2], prefix[3]};
    Q q = qmul(pre, quat_at(T, T.org_ts, T.org_q, T.org_n, qt));
    const double nn = sqrt(q.w * q.w + q.x * q.x + q.y * q.y + q.z * q.z);
    q = Q{q.w / nn, q.x / nn, q.y / nn, q.z / nn};
    double r[3][3] = {
        {1 - 2 * (q.y * q.y + q.z * q.z), 2 * (q.x * q.y - q.z * q.w), 2 * (q.x * q.z + q.y * q.w)},
        {2 * (q.x * q.y + q.z * q.w), 1 - 2 * (q.x * q.x + q.z * q.z), 2 * (q.y * q.z - q.x * q.w)},
        {2 * (q.x * q.z - q.y * q.w), 2 * (q.y * q.z + q.x * q.w), 1 - 2 * (q.x * q.x + q.y * q.y)}};
    if (F.video_rotation_deg != 0.0) {                                             // image_rotation * R
        const double a = F.video_rotation_deg * (3.14159265358979323846 / 180.0), ca = cos(a), sa = sin(a);
        for (int j = 0; j < 3; ++j) { const double r0 = r[0][j], r1 = r[1][j]; r[0][j] = ca * r0 - sa * r1; r[1][j] = sa * r0 + ca * r1; }
    }
    if (F.framebuffer_inverted) { r[0][2] *= -1.0; r[1][2] *= -1.0; r[2][0] *= -1.0; r[2][1] *= -1.0; }
    else { r[0][1] *= -1.0; r[0][2] *= -1.0; r[1][0] *= -1.0; r[2][0] *= -1.0; }
    // IBIS / OIS terms of this row (frame_transform.rs:270-289)
    float sx = 0.0f, sy = 0.0f, ra = 0.0f, ox = 0.0f, oy = 0.0f;
    if (S.ibis_n >= 0) {
        double y_sensor = ((double)y - 0.0) * ((S.crop_y + S.crop_h) - S.crop_y) / (S.height - 0.0) + S.crop_y;      // map_coord, util.rs:144-147
        if (F.framebuffer_inverted) y_sensor = S.sensor_h - y_sensor;
        double sv[3] = {0.0, 0.0, 0.0}, ov[3] = {0.0, 0.0, 0.0};
        if (!catmull_rom_at(S.ibis, S.ibis_n, y_sensor + S.offset, sv)) { sv[0] = 0.0; sv[1] = 0.0; sv[2] = 0.0; }
        if (!catmull_rom_at(S.ois, S.ois_n, y_sensor + S.offset, ov)) { ov[0] = 0.0; ov[1] = 0.0; ov[2] = 0.0; }
        const double rad = sv[2] / 1000.0 * (F.framebuffer_inverted ? -1.0 : 1.0);
        sx = (float)(sv[0] * S.scale_x); sy = (float)(sv[1] * S.scale_y);
        ra = (float)(rad * (3.14159265358979323846 / 180.0));      // to_radians()
        ox = (float)(ov[0] * S.scale_x); oy = (float)(ov[1] * S.scale_y);
    }
    if (F.suppress_rotation) {                                     // :291-296
        for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) r[i][j] = (i == j) ? 1.0 : 0.0;
        if (F.suppress_rotation == 2) { sx = 0.0f; sy = 0.0f; ra = 0.0f; ox = 0.0f; oy = 0.0f; }
    }
    double m[3][3];
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) m[i][j] = F.new_k[i * 3 + 0] * r[0][j] + F.new_k[i * 3 + 1] * r[1][j] + F.new_k[i * 3 + 2] * r[2][j];
    const double c00 = m[1][1] * m[2][2] - m[1][2] * m[2][1], c01 = m[1][2] * m[2][0] - m[1][0] * m[2][2], c02 = m[1][0] * m[2][1] - m[1][1] * m[2][0];
    const double det = m[0][0] * c00 + m[0][1] * c01 + m[0][2] * c02;
    const double id = 1.0 / det;
    float *o = out + (size_t)y * GFW_MAT_STRIDE;
    o[0] = (float)(c00 * id); o[1] = (float)((m[0][2] * m[2][1] - m[0][1] * m[2][2]) * id); o[2] = (float)((m[0][1] * m[1][2] - m[0][2] * m[1][1]) * id);
    o[3] = (float)(c01 * id); o[4] = (float)((m[0][0] * m[2][2] - m[0][2] * m[2][0]) * id); o[5] = (float)((m[0][2] * m[1][0] - m[0][0] * m[1][2]) * id);
    o[6] = (float)(c02 * id); o[7] = (float)((m[0][1] * m[2][0] - m[0][0] * m[2][1]) * id); o[8] = (float)((m[0][0] * m[1][1] - m[0][1] * m[1][0]) * id);
    o[9] = sx; o[10] = sy; o[11] = ra; o[12] = ox; o[13] = oy;
    // cos/sin of the roll as cpu_undistort.rs:159-160 evaluates them (host libm, restated in gfw_math.h), only for rows with data
    if (sx != 0.0f || sy != 0.0f || ra != 0.0f || ox != 0.0f || oy != 0.0f) { o[14] = gfw_cosf(-ra); o[15] = gfw_sinf(-ra); }
    else { o[14] = 1.0f; o[15] = 0.0f; }
}

}  // namespace

hipError_t gfw_launch_build_matrices(const GfwTracks &T, const gfw_frame_timing *d_timings, int frames, int max_rows, double *prefix_scratch,
                                     float *out, size_t table_floats, hipStream_t s, const GfwStab *stab) {
    if (frames <= 0 || max_rows <= 0) return hipSuccess;
    GfwStab S;
    if (stab) S = *stab; else { S = GfwStab{0, 0, 0, 0, 0, 0, 0, nullptr, nullptr, -1, -1}; }
    hipLaunchKernelGGL(gfw_build_prefix_kernel, dim3((frames + 63) / 64), dim3(64), 0, s, T, d_timings, frames, prefix_scratch);
    hipLaunchKernelGGL(gfw_build_matrices_kernel, dim3((max_rows + 63) / 64, frames), dim3(64), 0, s, T, d_timings, (const double *)prefix_scratch, out, table_floats, S);
    return hipGetLastError();
}
