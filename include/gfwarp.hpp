// gfwarp.hpp — C++ host-side mirror of gyroflow-core's operator surface for the warp path, on top of the C ABI
// (include/gfwarp.h).  Header-only, C++17, no HIP or torch types.
//
// The reference's host side is compiled Rust; its toolchain is not available here, so the host logic above the C ABI is
// restated in C++ with the reference's names, argument meaning and error behaviour:
//
//   gyroflow::Stabilization                src/core/stabilization/mod.rs:169-192
//     ::init_size                          mod.rs:375
//     ::get_kernel_flags                   mod.rs:226-251
//     ::get_frame_transform_at<T>          mod.rs:253-326   (completes KernelParams from the buffers and the pixel type)
//     ::ensure_ready_for_processing<T>     mod.rs:567-611   (backend object cache, LRU(15) as mod.rs:59-66)
//     ::process_pixels<T>                  mod.rs:612-725   (validation, then the backend arm; there is no CPU arm here)
//   gyroflow::BufferDescription / Buffers  src/core/gpu/mod.rs:17-28
//   gyroflow::FrameTransform               src/core/stabilization/frame_transform.rs:12-19
//   gyroflow::GyroflowCoreError            src/core/lib.rs:2099-2141
//   gyroflow::{Luma8, Luma16, ...}         src/core/stabilization/pixel_formats.rs (PixelType implementors)
//
// `FrameTransform::at_timestamp` itself (quaternions -> per-row matrices) is input here: the caller provides
// `matrices`, or builds them on the device with gfw_build_matrices / gfw_build_matrices_batch.
#pragma once
#include <array>
#include <cstdint>
#include <cstring>
#include <limits>
#include <list>
#include <memory>
#include <optional>
#include <stdexcept>
#include <string>
#include <tuple>
#include <utility>
#include <vector>

#include "gfwarp.h"

namespace gyroflow {

using KernelParams = gfw_kernel_params;
static_assert(sizeof(KernelParams) == 368, "KernelParams is #[repr(C, packed(4))], 368 bytes (mod.rs:101-150)");

enum class Interpolation : int32_t {                 // mod.rs:25-34
    Bilinear = 2, Bicubic = 4, Lanczos4 = 8, RobidouxSharp = 10, Robidoux = 11, Mitchell = 12, CatmullRom = 13
};

// GyroflowCoreError (lib.rs:2099-2141): the variants the pixel path can raise
struct GyroflowCoreError : std::runtime_error {
    enum Kind { SizeTooSmall, SizeMismatch, InvalidStride, NoStabilizationData, InputBufferEmpty, OutputBufferEmpty, Unknown };
    Kind kind;
    GyroflowCoreError(Kind k, const std::string &detail = "") : std::runtime_error(name(k) + (detail.empty() ? "" : ": " + detail)), kind(k) {}
    static std::string name(Kind k) {
        static const char *n[] = {"SizeTooSmall", "SizeMismatch", "InvalidStride", "NoStabilizationData", "InputBufferEmpty", "OutputBufferEmpty", "Unknown"};
        return n[k];
    }
    static Kind from_code(int rc) {
        switch (rc) {
        case GFW_ERR_SIZE_TOO_SMALL: return SizeTooSmall;
        case GFW_ERR_SIZE_MISMATCH: return SizeMismatch;
        case GFW_ERR_INVALID_STRIDE: return InvalidStride;
        case GFW_ERR_NO_STABILIZATION_DATA: return NoStabilizationData;
        case GFW_ERR_INPUT_BUFFER_EMPTY: return InputBufferEmpty;
        case GFW_ERR_OUTPUT_BUFFER_EMPTY: return OutputBufferEmpty;
        default: return Unknown;
        }
    }
};

// ---- PixelType implementors (pixel_formats.rs:48-60, :62-302): id, scalar type, element count, default_max_value ----
template <int Id, typename Scalar_, int Count_, bool HasMax, int MaxValue> struct PixelTypeT {
    static constexpr int ID = Id;
    using Scalar = Scalar_;
    static constexpr int COUNT = Count_;
    static constexpr int SCALAR_BYTES = (int)sizeof(Scalar_);
    static constexpr int BYTES = COUNT * SCALAR_BYTES;
    static std::optional<float> default_max_value() { return HasMax ? std::optional<float>((float)MaxValue) : std::nullopt; }
};
using Luma8   = PixelTypeT<GFW_PIX_LUMA8,   uint8_t,  1, true, 255>;
using Luma16  = PixelTypeT<GFW_PIX_LUMA16,  uint16_t, 1, true, 65535>;
using RGB8    = PixelTypeT<GFW_PIX_RGB8,    uint8_t,  3, true, 255>;
using RGBA8   = PixelTypeT<GFW_PIX_RGBA8,   uint8_t,  4, true, 255>;
using BGRA8   = PixelTypeT<GFW_PIX_BGRA8,   uint8_t,  4, true, 255>;
using RGB16   = PixelTypeT<GFW_PIX_RGB16,   uint16_t, 3, true, 65535>;
using RGBA16  = PixelTypeT<GFW_PIX_RGBA16,  uint16_t, 4, true, 65535>;
using AYUV16  = PixelTypeT<GFW_PIX_AYUV16,  uint16_t, 4, true, 65535>;
using RGBAf   = PixelTypeT<GFW_PIX_RGBAF,   float,    4, false, 0>;
using RGBAf16 = PixelTypeT<GFW_PIX_RGBAF16, uint16_t, 4, false, 0>;      // half::f16 storage
using R32f    = PixelTypeT<GFW_PIX_R32F,    float,    1, false, 0>;
using UV8     = PixelTypeT<GFW_PIX_UV8,     uint8_t,  2, true, 255>;
using UV16    = PixelTypeT<GFW_PIX_UV16,    uint16_t, 2, true, 65535>;

// ---- Buffers (gpu/mod.rs:17-71) ----
struct BufferSource {
    enum Kind { None, Cpu, HipDevice } kind = None;   // Cpu{buffer: &mut [u8]}; HipDevice = the CUDABuffer{buffer} analogue
    void *data = nullptr;
    size_t len = 0;
    static BufferSource cpu(void *p, size_t n) { return {Cpu, p, n}; }
    static BufferSource hip_device(void *p, size_t n) { return {HipDevice, p, n}; }
};
struct BufferDescription {
    std::tuple<size_t, size_t, size_t> size{0, 0, 0};                     // (width, height, stride in bytes)
    std::optional<std::tuple<size_t, size_t, size_t, size_t>> rect;       // (x, y, w, h)
    std::optional<float> rotation;
    BufferSource data;
    bool texture_copy = false;
    gfw_buffer_desc to_abi() const {
        gfw_buffer_desc d;
        std::memset(&d, 0, sizeof(d));
        d.width = (int32_t)std::get<0>(size); d.height = (int32_t)std::get<1>(size); d.stride = (int32_t)std::get<2>(size);
        if (rect) { d.has_rect = 1; d.rect[0] = (int32_t)std::get<0>(*rect); d.rect[1] = (int32_t)std::get<1>(*rect); d.rect[2] = (int32_t)std::get<2>(*rect); d.rect[3] = (int32_t)std::get<3>(*rect); }
        if (rotation) { d.has_rotation = 1; d.rotation = *rotation; }
        d.kind = data.kind == BufferSource::Cpu ? GFW_BUF_HOST : data.kind == BufferSource::HipDevice ? GFW_BUF_HIP_DEVICE : GFW_BUF_NONE;
        d.texture_copy = texture_copy ? 1 : 0;
        d.data = data.data; d.len = data.len;
        return d;
    }
};
struct Buffers {
    BufferDescription input, output;
    gfw_buffers to_abi() const { gfw_buffers b; b.input = input.to_abi(); b.output = output.to_abi(); return b; }
    // get_checksum (gpu/mod.rs:110-135): what makes a backend object reusable for another call
    uint64_t get_checksum() const {
        uint64_t h = 1469598103934665603ull;
        auto mix = [&h](uint64_t v) { h ^= v; h *= 1099511628211ull; };
        for (const BufferDescription *d : {&input, &output}) {
            mix(std::get<0>(d->size)); mix(std::get<1>(d->size)); mix(std::get<2>(d->size));
            mix(d->rect ? 1 : 0);
            if (d->rect) { mix(std::get<0>(*d->rect)); mix(std::get<1>(*d->rect)); mix(std::get<2>(*d->rect)); mix(std::get<3>(*d->rect)); }
            mix((uint64_t)d->data.kind);
        }
        return h;
    }
};

// ---- FrameTransform (frame_transform.rs:12-19) ----
struct FrameTransform {
    std::vector<std::array<float, 14>> matrices;
    KernelParams kernel_params{};
    double fov = 1.0, minimal_fov = 1.0;
    std::optional<double> focal_length;
    std::vector<float> mesh_data;
};
struct ProcessedInfo {                                                    // mod.rs:194-202
    double fov, minimal_fov;
    std::optional<double> focal_length;
    std::string backend;
};

// The slice of ComputeParams (compute_params.rs:71-138) that get_kernel_flags / the backend key consult
struct ComputeParams {
    int distortion_model = GFW_MODEL_OPENCV_FISHEYE;     // GFW_MODEL_* of params.distortion_model
    int digital_lens = GFW_MODEL_NONE;                   // params.digital_lens (None = GFW_MODEL_NONE)
    bool horizontal_rs = false;                          // frame_readout_direction.is_horizontal()
    bool framebuffer_inverted = false;
    float light_refraction_coefficient = 1.0f;
    std::array<float, 4> background{0.0f, 0.0f, 0.0f, 0.0f};
};

class Stabilization {
  public:
    std::pair<size_t, size_t> size{0, 0}, output_size{0, 0};
    Interpolation interpolation = Interpolation::Bilinear;
    int32_t kernel_flags = 0;
    ComputeParams compute_params;
    std::string initialized_backend;                                      // "" until a backend object exists

    ~Stabilization() { for (auto &e : backends_) gfw_destroy(e.second); }
    Stabilization() = default;
    Stabilization(const Stabilization &) = delete;
    Stabilization &operator=(const Stabilization &) = delete;

    void set_compute_params(const ComputeParams &p) { compute_params = p; }                        // mod.rs:204
    void init_size(std::pair<size_t, size_t> s, std::pair<size_t, size_t> out) { initialized_backend.clear(); size = s; output_size = out; }   // mod.rs:375

    static std::tuple<size_t, size_t, size_t, size_t> get_rect(const BufferDescription &d) {       // mod.rs:209-224
        if (d.rect) return *d.rect;
        return {0, 0, std::get<0>(d.size), std::get<1>(d.size)};
    }
    int32_t get_kernel_flags(const Buffers &b) const {                                             // mod.rs:226-251
        int32_t f = kernel_flags;
        auto set = [&f](int32_t bit, bool on) { f = on ? (f | bit) : (f & ~bit); };
        set(GFW_FLAG_HAS_DIGITAL_LENS, compute_params.digital_lens != GFW_MODEL_NONE);
        set(GFW_FLAG_HORIZONTAL_RS, compute_params.horizontal_rs);
        set(GFW_FLAG_HAS_SOURCE_RECT, b.input.rect.has_value() || size != std::make_pair(std::get<0>(b.input.size), std::get<1>(b.input.size)));
        set(GFW_FLAG_HAS_OUTPUT_RECT, b.output.rect.has_value() || output_size != std::make_pair(std::get<0>(b.output.size), std::get<1>(b.output.size)));
        set(GFW_FLAG_FRAMEBUFFER_INVERTED, compute_params.framebuffer_inverted);
        set(GFW_FLAG_ANY_UNDERWATER, compute_params.light_refraction_coefficient != 1.0f && compute_params.light_refraction_coefficient > 0.0f);
        return f;
    }

    // mod.rs:253-326: complete the per-frame KernelParams (lens, fov, matrix_count, ... already filled by
    // FrameTransform::at_timestamp, frame_transform.rs:322-340) from the buffers and the pixel type.
    template <typename T>
    FrameTransform get_frame_transform_at(FrameTransform transform, const Buffers &b) const {
        KernelParams &p = transform.kernel_params;
        p.pixel_value_limit = T::default_max_value().value_or(std::numeric_limits<float>::max());
        p.max_pixel_value = T::default_max_value().value_or(1.0f);
        p.interpolation = (int32_t)interpolation;                                                  // before the EWA test below
        p.width = (int32_t)size.first; p.height = (int32_t)size.second;
        p.output_width = (int32_t)output_size.first; p.output_height = (int32_t)output_size.second;
        for (int i = 0; i < 4; ++i) p.background[i] = compute_params.background[i];
        p.bytes_per_pixel = T::BYTES;
        p.pix_element_count = T::COUNT;
        p.canvas_scale = 1.0f;
        p.flags = get_kernel_flags(b);
        p.stride = (int32_t)std::get<2>(b.input.size);
        p.output_stride = (int32_t)std::get<2>(b.output.size);
        if (p.interpolation > 8) {                                                                 // mod.rs:279-295 (Keys cubic family)
            float B = 0.0f, C = 0.5f;
            switch (interpolation) {
            case Interpolation::RobidouxSharp: B = 0.2620145f; C = 0.3689927f; break;
            case Interpolation::Robidoux:      B = 0.3782157f; C = 0.3108921f; break;
            case Interpolation::Mitchell:      B = 0.3333333f; C = 0.3333333f; break;
            default:                           B = 0.0f;       C = 0.5f;       break;              // CatmullRom
            }
            p.ewa_coeffs_p[0] = (6.0f - 2.0f * B) / 6.0f;
            p.ewa_coeffs_p[1] = 0.0f;
            p.ewa_coeffs_p[2] = (-18.0f + 12.0f * B + 6.0f * C) / 6.0f;
            p.ewa_coeffs_p[3] = (12.0f - 9.0f * B - 6.0f * C) / 6.0f;
            p.ewa_coeffs_q[0] = (8.0f * B + 24.0f * C) / 6.0f;
            p.ewa_coeffs_q[1] = (-12.0f * B - 48.0f * C) / 6.0f;
            p.ewa_coeffs_q[2] = (6.0f * B + 30.0f * C) / 6.0f;
            p.ewa_coeffs_q[3] = (-1.0f * B - 6.0f * C) / 6.0f;
        }
        p.safe_area_rect[0] = 0.0f; p.safe_area_rect[1] = 0.0f;
        p.safe_area_rect[2] = (float)output_size.first; p.safe_area_rect[3] = (float)output_size.second;
        if (b.input.rotation) p.input_rotation = *b.input.rotation;
        if (b.output.rotation) p.output_rotation = *b.output.rotation;
        const auto sr = get_rect(b.input), orr = get_rect(b.output);
        p.source_rect[0] = (int32_t)std::get<0>(sr); p.source_rect[1] = (int32_t)std::get<1>(sr); p.source_rect[2] = (int32_t)std::get<2>(sr); p.source_rect[3] = (int32_t)std::get<3>(sr);
        p.output_rect[0] = (int32_t)std::get<0>(orr); p.output_rect[1] = (int32_t)std::get<1>(orr); p.output_rect[2] = (int32_t)std::get<2>(orr); p.output_rect[3] = (int32_t)std::get<3>(orr);
        return transform;
    }

    // mod.rs:355-373: what a backend object is keyed by (FILL_WITH_BACKGROUND stays a run-time flag, opencl.rs:209)
    uint64_t get_current_checksum(const Buffers &b, int pixel_type) const {
        uint64_t h = b.get_checksum();
        auto mix = [&h](uint64_t v) { h ^= v; h *= 1099511628211ull; };
        mix((uint64_t)compute_params.distortion_model); mix((uint64_t)compute_params.digital_lens);
        mix((uint64_t)(int32_t)interpolation);
        mix((uint64_t)(get_kernel_flags(b) & ~GFW_FLAG_FILL_WITH_BACKGROUND));
        mix(size.first); mix(size.second); mix(output_size.first); mix(output_size.second);
        mix((uint64_t)pixel_type);
        return h;
    }

    // mod.rs:567-611 (init_backends :467-565 restricted to the HIP arm): create or reuse the backend object
    template <typename T>
    gfw_ctx *ensure_ready_for_processing(const KernelParams &kp, const Buffers &b) {
        const uint64_t key = get_current_checksum(b, T::ID);
        for (auto it = backends_.begin(); it != backends_.end(); ++it)
            if (it->first == key) { backends_.splice(backends_.begin(), backends_, it); return backends_.front().second; }
        const gfw_buffers ab = b.to_abi();
        gfw_ctx *ctx = gfw_create(&kp, T::ID, compute_params.distortion_model, compute_params.digital_lens, &ab, 0);
        if (!ctx) throw GyroflowCoreError(GyroflowCoreError::Unknown, std::string("backend initialisation failed: ") + gfw_last_error());
        backends_.emplace_front(key, ctx);
        while (backends_.size() > 15) { gfw_destroy(backends_.back().second); backends_.pop_back(); }   // LRU(15), mod.rs:59-66
        char info[256] = {0};
        initialized_backend = gfw_get_info(info, sizeof(info)) >= 0 ? std::string("HIP: ") + info : "HIP";
        return ctx;
    }

    // mod.rs:612-725.  `frame_transform` = None -> NoStabilizationData (this mirror keeps no stab_data cache).
    template <typename T>
    ProcessedInfo process_pixels(int64_t timestamp_us, std::optional<size_t> /*frame*/, Buffers &buffers, const FrameTransform *frame_transform) {
        if (std::get<1>(buffers.input.size) < 4 || std::get<1>(buffers.output.size) < 4) throw GyroflowCoreError(GyroflowCoreError::SizeTooSmall);   // :613
        if (!frame_transform) throw GyroflowCoreError(GyroflowCoreError::NoStabilizationData, std::to_string(timestamp_us));                       // :721
        const FrameTransform &itm = *frame_transform;
        const KernelParams &kp = itm.kernel_params;
        if (size != std::make_pair((size_t)kp.width, (size_t)kp.height)) throw GyroflowCoreError(GyroflowCoreError::SizeMismatch);                  // :636
        if (output_size != std::make_pair((size_t)kp.output_width, (size_t)kp.output_height)) throw GyroflowCoreError(GyroflowCoreError::SizeMismatch);
        if ((int64_t)std::get<0>(buffers.input.size) > kp.stride) throw GyroflowCoreError(GyroflowCoreError::InvalidStride);                         // :639
        if ((int64_t)std::get<0>(buffers.output.size) > kp.output_stride) throw GyroflowCoreError(GyroflowCoreError::InvalidStride);
        if (buffers.input.data.kind == BufferSource::None || buffers.input.data.len == 0) throw GyroflowCoreError(GyroflowCoreError::InputBufferEmpty);     // lib.rs:890
        if (buffers.output.data.kind == BufferSource::None || buffers.output.data.len == 0) throw GyroflowCoreError(GyroflowCoreError::OutputBufferEmpty);  // lib.rs:891
        gfw_ctx *ctx = ensure_ready_for_processing<T>(kp, buffers);
        const gfw_buffers ab = buffers.to_abi();
        const int rc = gfw_undistort_image(ctx, &ab, &kp, itm.matrices.empty() ? nullptr : itm.matrices[0].data(), (int)itm.matrices.size(), nullptr, 0,
                                           itm.mesh_data.empty() ? nullptr : itm.mesh_data.data(), itm.mesh_data.size());
        if (rc != GFW_OK) throw GyroflowCoreError(GyroflowCoreError::from_code(rc), gfw_last_error());
        return ProcessedInfo{itm.fov, itm.minimal_fov, itm.focal_length, std::string("HIP:") + gfw_last_backend(ctx)};
    }

  private:
    std::list<std::pair<uint64_t, gfw_ctx *>> backends_;
};

}  // namespace gyroflow
