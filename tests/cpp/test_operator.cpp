// C++ use of the operator mirror (include/gfwarp.hpp), written the way the reference's render loop uses Stabilization
// (src/rendering/mod.rs:494-542): one Stabilization per plane, get_frame_transform_at::<T>, process_pixels::<T>.
//
//   test_operator validate        error behaviour of process_pixels (mod.rs:612-640, lib.rs:890-891); needs no GPU
//   test_operator warp <oracle>   Luma16 plane, opencv_fisheye, per-row rolling-shutter matrices through the HIP backend,
//                                 compared bit-exactly with the oracle (dlopen'ed: the checker, test infrastructure)
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <dlfcn.h>
#include <string>
#include <vector>

#include "gfwarp.hpp"

using namespace gyroflow;

#define CHECK(cond) do { if (!(cond)) { std::printf("FAILED %s:%d: %s\n", __FILE__, __LINE__, #cond); std::exit(1); } } while (0)

static FrameTransform synthetic_transform(int w, int h, int rows) {
    // what FrameTransform::at_timestamp fills (frame_transform.rs:322-340) for a GoPro-style fisheye and a small,
    // row-dependent rotation about the optical axis + pitch (rolling shutter): matrices = inv(K * R_row)
    FrameTransform t;
    KernelParams &p = t.kernel_params;
    std::memset(&p, 0, sizeof(p));
    const double fx = 0.47 * w, cx = w / 2.0, cy = h / 2.0;
    p.f[0] = p.f[1] = (float)fx; p.c[0] = (float)cx; p.c[1] = (float)cy;
    p.k[0] = 0.045f; p.k[1] = 0.02f; p.k[2] = -0.02f; p.k[3] = 0.006f;
    p.fov = 1.0f; p.lens_correction_amount = 1.0f; p.input_vertical_stretch = 1.0f; p.input_horizontal_stretch = 1.0f;
    p.light_refraction_coefficient = 1.0f;
    p.matrix_count = rows;
    t.matrices.resize(rows);
    for (int y = 0; y < rows; ++y) {
        const double roll = 0.02 * std::sin(y * 0.01), pitch = 0.015 * (double)y / rows - 0.007;
        const double cr = std::cos(roll), sr = std::sin(roll), cp = std::cos(pitch), sp = std::sin(pitch);
        const double R[3][3] = {{cr, -sr, 0}, {sr * cp, cr * cp, -sp}, {sr * sp, cr * sp, cp}};
        // inv(K R) = R^T K^-1
        const double Ki[3][3] = {{1 / fx, 0, -cx / fx}, {0, 1 / fx, -cy / fx}, {0, 0, 1}};
        for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) {
            double s = 0; for (int k = 0; k < 3; ++k) s += R[k][i] * Ki[k][j];
            t.matrices[y][i * 3 + j] = (float)s;
        }
        for (int i = 9; i < 14; ++i) t.matrices[y][i] = 0.0f;
    }
    return t;
}

template <typename S> static std::vector<uint8_t> pattern(int w, int h, size_t stride) {
    std::vector<uint8_t> buf(stride * h, 0);
    uint64_t st = 0x9E3779B97F4A7C15ull;
    for (int y = 0; y < h; ++y) for (int x = 0; x < w; ++x) {
        st += 0x9E3779B97F4A7C15ull; uint64_t z = st; z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; z ^= z >> 31;
        const double smooth = 0.5 + 0.5 * std::sin(x * 0.07) * std::cos(y * 0.05);
        const S v = (S)((double)std::numeric_limits<S>::max() * (0.7 * smooth + 0.3 * (double)(z & 0xffff) / 65535.0));
        std::memcpy(&buf[y * stride + x * sizeof(S)], &v, sizeof(S));
    }
    return buf;
}

static int run_validate() {
    const int w = 64, h = 48;
    Stabilization stab;
    stab.init_size({w, h}, {w, h});
    std::vector<uint8_t> src(w * 2 * h), dst(w * 2 * h);
    Buffers b;
    b.input.size = {w, h, w * 2}; b.input.data = BufferSource::cpu(src.data(), src.size());
    b.output.size = {w, h, w * 2}; b.output.data = BufferSource::cpu(dst.data(), dst.size());
    FrameTransform t = stab.get_frame_transform_at<Luma16>(synthetic_transform(w, h, h), b);
    CHECK(t.kernel_params.bytes_per_pixel == 2 && t.kernel_params.pix_element_count == 1);
    CHECK(t.kernel_params.max_pixel_value == 65535.0f && t.kernel_params.pixel_value_limit == 65535.0f);
    CHECK(t.kernel_params.source_rect[2] == w && t.kernel_params.output_rect[3] == h && t.kernel_params.stride == w * 2);
    CHECK((t.kernel_params.flags & (GFW_FLAG_HAS_SOURCE_RECT | GFW_FLAG_HAS_OUTPUT_RECT)) == 0);
    auto expect = [&](GyroflowCoreError::Kind k, Buffers &bb, const FrameTransform *ft) {
        try { stab.process_pixels<Luma16>(1000, std::nullopt, bb, ft); } catch (const GyroflowCoreError &e) { CHECK(e.kind == k); return; }
        CHECK(!"expected GyroflowCoreError");
    };
    expect(GyroflowCoreError::NoStabilizationData, b, nullptr);
    { Buffers s = b; s.input.size = {w, 3, w * 2}; expect(GyroflowCoreError::SizeTooSmall, s, &t); }
    { FrameTransform u = t; u.kernel_params.width = w + 2; expect(GyroflowCoreError::SizeMismatch, b, &u); }
    { FrameTransform u = t; u.kernel_params.output_height = h - 2; expect(GyroflowCoreError::SizeMismatch, b, &u); }
    { FrameTransform u = t; u.kernel_params.stride = w - 1; expect(GyroflowCoreError::InvalidStride, b, &u); }
    { Buffers s = b; s.input.data = BufferSource{}; expect(GyroflowCoreError::InputBufferEmpty, s, &t); }
    { Buffers s = b; s.output.data = BufferSource::cpu(dst.data(), 0); expect(GyroflowCoreError::OutputBufferEmpty, s, &t); }
    // sub-rectangle buffers set the rect flags (mod.rs:238-241)
    { Buffers s = b; s.input.rect = std::make_tuple((size_t)4, (size_t)4, (size_t)32, (size_t)32);
      CHECK(stab.get_kernel_flags(s) & GFW_FLAG_HAS_SOURCE_RECT);
      FrameTransform u = stab.get_frame_transform_at<Luma16>(synthetic_transform(w, h, h), s);
      CHECK(u.kernel_params.source_rect[0] == 4 && u.kernel_params.source_rect[2] == 32); }
    // EWA coefficients (mod.rs:279-295)
    { stab.interpolation = Interpolation::Mitchell;
      FrameTransform u = stab.get_frame_transform_at<Luma16>(synthetic_transform(w, h, h), b);
      CHECK(std::fabs(u.kernel_params.ewa_coeffs_p[0] - (6.0f - 2.0f * 0.3333333f) / 6.0f) == 0.0f && u.kernel_params.interpolation == 12);
      stab.interpolation = Interpolation::Bilinear; }
    // with every check passed the backend arm is next: on a box without a GPU it must fail loudly, never fall back
    if (gfw_list_devices(nullptr, 0) <= 0) {
        try { stab.process_pixels<Luma16>(1000, std::nullopt, b, &t); CHECK(!"expected a loud failure without a device"); }
        catch (const GyroflowCoreError &e) { CHECK(e.kind == GyroflowCoreError::Unknown); std::printf("no device: %s\n", e.what()); }
    }
    std::printf("validate ok\n");
    return 0;
}

typedef int (*oracle_fn)(const gfw_buffers *, const gfw_kernel_params *, int, int, int, const float *, const float *, size_t, int);

static int run_warp(const char *oracle_path) {
    void *so = dlopen(oracle_path, RTLD_NOW);
    CHECK(so != nullptr);
    oracle_fn oracle = (oracle_fn)dlsym(so, "gfw_oracle_undistort_image");
    CHECK(oracle != nullptr);
    const int w = 320, h = 180;
    const size_t stride = 768;                       // padded rows: padding bytes must stay untouched
    Stabilization stab;
    stab.init_size({w, h}, {w, h});
    std::vector<uint8_t> src = pattern<uint16_t>(w, h, stride), dst(stride * h, 0x5A), ref(stride * h, 0x5A);
    Buffers b;
    b.input.size = {w, h, stride}; b.input.data = BufferSource::cpu(src.data(), src.size());
    b.output.size = {w, h, stride}; b.output.data = BufferSource::cpu(dst.data(), dst.size());
    for (Interpolation interp : {Interpolation::Bilinear, Interpolation::Lanczos4, Interpolation::Robidoux}) {
        stab.interpolation = interp;
        const FrameTransform t = stab.get_frame_transform_at<Luma16>(synthetic_transform(w, h, h), b);
        const ProcessedInfo info = stab.process_pixels<Luma16>(33333, std::nullopt, b, &t);
        gfw_buffers rb = b.to_abi();
        rb.output.data = ref.data();
        CHECK(oracle(&rb, &t.kernel_params, Luma16::ID, GFW_MODEL_OPENCV_FISHEYE, GFW_MODEL_NONE, t.matrices[0].data(), nullptr, 0, 0) == 1);
        size_t diff = 0, touched = 0;
        for (size_t i = 0; i < dst.size(); ++i) { diff += dst[i] != ref[i]; touched += ref[i] != 0x5A; }
        std::printf("interpolation %d: backend %s, %zu differing bytes, %zu bytes written\n", (int)interp, info.backend.c_str(), diff, touched);
        CHECK(diff == 0 && touched > (size_t)w * h);
        for (int y = 0; y < h; ++y) for (size_t x = (size_t)w * 2; x < stride; ++x) CHECK(dst[y * stride + x] == 0x5A);
    }
    CHECK(!stab.initialized_backend.empty());
    std::printf("warp ok (%s)\n", stab.initialized_backend.c_str());
    return 0;
}

int main(int argc, char **argv) {
    if (argc >= 2 && std::string(argv[1]) == "validate") return run_validate();
    if (argc >= 3 && std::string(argv[1]) == "warp") return run_warp(argv[2]);
    std::printf("usage: test_operator validate | warp <liboracle.so>\n");
    return 2;
}
