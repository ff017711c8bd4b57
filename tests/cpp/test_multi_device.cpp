// N renders in ONE process, through the C ABI (round-5 verdict, next #9): the reference's closest shape to multi-GPU is `--parallel-renders` — several renders
// side by side in one process (src/cli.rs:46-48, render_queue.rs:677), each with its own Stabilization / backend objects on its own thread
// (stabilization/mod.rs:59-66: backends live in thread-local caches).  Here: T threads, thread t calls gfw_set_device(t mod ndev) (the device choice is per
// thread, like hipSetDevice), creates its own context and warps the frames t, t + T, t + 2T, ... of one synthetic clip (planar 4:2:2, 16-bit: three planes per
// frame, per-row rolling-shutter matrices that differ from frame to frame), with gfw_set_frame_checksums taking each frame's checksum where the pixels leave.
// The words must equal those of the same clip warped by ONE thread on device 0, frame for frame — whichever kernel served a frame (ahead of time for the
// first frames of a context, the run-time specialised one once its background build lands: GFW_OPT_JIT = 1, the default).  On the 1-GPU box ndev = 1 and the
// four threads share the device; on an 8-GPU node this is the C-level twin of `bench.py --c5 --gpus 8` (frame k -> thread k mod T = SURVEY.md 8e's round-robin).
//
//   test_multi_device [threads = 4] [frames = 36]          (built by __graft_entry__.build_test_helpers with hipcc: it allocates device memory itself)
#include <hip/hip_runtime.h>
#include <unistd.h>
#include <csignal>
#include <atomic>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "gfwarp.hpp"

using namespace gyroflow;

// (a failing check on one thread leaves at once, through _exit: exit() would run the process's exit handlers — the library joins its build threads there — beside
//  threads that are still inside the HIP runtime, and a failure would show as a hang with nothing printed)
#define CHECK(cond) do { if (!(cond)) { std::fprintf(stderr, "FAILED %s:%d: %s (%s)\n", __FILE__, __LINE__, #cond, gfw_last_error()); std::fflush(stderr); _exit(1); } } while (0)
#define HIPOK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "FAILED %s:%d: %s -> %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_)); std::fflush(stderr); _exit(1); } } while (0)

// progress goes to stderr, unbuffered, so that a run that stops shows WHERE (a 4-thread run once hung beside three other GPU processes and said nothing: r06_c)
#define MARK(...) do { std::fprintf(stderr, __VA_ARGS__); std::fflush(stderr); } while (0)
static std::atomic<int> g_stage{0};
static void on_alarm(int) { char b[96]; const int n = std::snprintf(b, sizeof(b), "\nFAILED: watchdog, no progress for 120 s (stage %d)\n", g_stage.load()); (void)!write(2, b, (size_t)n); _exit(4); }

static const int W = 640, H = 360, CW = W / 2;                    // 4:2:2: chroma planes half as wide, as tall
static const size_t YS = 1536, CS = 768;                          // row pitches (padded)

// what FrameTransform::at_timestamp fills (frame_transform.rs:322-340) for a GoPro-style fisheye; frame f's per-row matrices = inv(K R_row(f))
static FrameTransform transform_of(int f) {
    FrameTransform t;
    KernelParams &p = t.kernel_params;
    std::memset(&p, 0, sizeof(p));
    const double fx = 0.47 * W, cx = W / 2.0, cy = H / 2.0;
    p.f[0] = p.f[1] = (float)fx; p.c[0] = (float)cx; p.c[1] = (float)cy;
    p.k[0] = 0.045f; p.k[1] = 0.02f; p.k[2] = -0.02f; p.k[3] = 0.006f;
    p.fov = 1.0f; p.lens_correction_amount = 1.0f; p.input_vertical_stretch = 1.0f; p.input_horizontal_stretch = 1.0f;
    p.light_refraction_coefficient = 1.0f;
    p.matrix_count = H;
    t.matrices.resize(H);
    for (int y = 0; y < H; ++y) {
        const double roll = 0.02 * std::sin(y * 0.01 + f * 0.37), pitch = 0.015 * (double)y / H - 0.007 + 0.004 * std::cos(f * 0.21);
        const double cr = std::cos(roll), sr = std::sin(roll), cp = std::cos(pitch), sp = std::sin(pitch);
        const double R[3][3] = {{cr, -sr, 0}, {sr * cp, cr * cp, -sp}, {sr * sp, cr * sp, cp}};
        const double Ki[3][3] = {{1 / fx, 0, -cx / fx}, {0, 1 / fx, -cy / fx}, {0, 0, 1}};
        for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) {
            double s = 0; for (int k = 0; k < 3; ++k) s += R[k][i] * Ki[k][j];
            t.matrices[y][i * 3 + j] = (float)s;
        }
        for (int i = 9; i < 14; ++i) t.matrices[y][i] = 0.0f;
    }
    return t;
}

static std::vector<uint8_t> pattern(int w, int h, size_t stride, uint64_t seed) {
    std::vector<uint8_t> buf(stride * h, 0);
    uint64_t st = 0x9E3779B97F4A7C15ull * (seed + 1);
    for (int y = 0; y < h; ++y) for (int x = 0; x < w; ++x) {
        st += 0x9E3779B97F4A7C15ull; uint64_t z = st; z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; z ^= z >> 31;
        const double smooth = 0.5 + 0.5 * std::sin(x * 0.07 + seed) * std::cos(y * 0.05);
        const uint16_t v = (uint16_t)(65535.0 * (0.7 * smooth + 0.3 * (double)(z & 0xffff) / 65535.0));
        std::memcpy(&buf[y * stride + x * 2], &v, 2);
    }
    return buf;
}

// One render: its own device buffers, its own context on the calling thread's device, the frames `first, first + step, ...` of the clip; sums[k] <- frame k's checksum.
static void render(int device, int first, int step, int n_frames, const std::vector<uint8_t> src[3], std::vector<unsigned long long> *sums, std::string *backends) {
    CHECK(gfw_set_device(device) == GFW_OK);
    HIPOK(hipSetDevice(device));
    MARK("[render first %d step %d on device %d: start]\n", first, step, device);
    const int pw[3] = {W, CW, CW};
    const size_t ps[3] = {YS, CS, CS};
    uint8_t *d_src[3], *d_dst[3];
    for (int i = 0; i < 3; ++i) {
        HIPOK(hipMalloc((void **)&d_src[i], ps[i] * H)); HIPOK(hipMalloc((void **)&d_dst[i], ps[i] * H));
        HIPOK(hipMemcpy(d_src[i], src[i].data(), ps[i] * H, hipMemcpyHostToDevice));
    }
    int mine = 0;
    for (int k = first; k < n_frames; k += step) ++mine;
    unsigned long long *d_sums;
    HIPOK(hipMalloc((void **)&d_sums, sizeof(unsigned long long) * (size_t)(mine > 0 ? mine : 1)));
    HIPOK(hipMemset(d_sums, 0, sizeof(unsigned long long) * (size_t)(mine > 0 ? mine : 1)));
    HIPOK(hipDeviceSynchronize());

    // the render loop's per-plane Stabilization objects (rendering/mod.rs:494-545) complete each plane's KernelParams; the frame then leaves as ONE C-ABI call
    Stabilization stab[3];
    Buffers b[3];
    for (int i = 0; i < 3; ++i) {
        stab[i].init_size({W, H}, {W, H});                                  // every plane's Stabilization knows the FULL frame size (rendering/mod.rs:514)
        b[i].input.size = {(size_t)pw[i], (size_t)H, ps[i]};  b[i].input.data = BufferSource::hip_device(d_src[i], ps[i] * H);
        b[i].output.size = {(size_t)pw[i], (size_t)H, ps[i]}; b[i].output.data = BufferSource::hip_device(d_dst[i], ps[i] * H);
    }
    gfw_ctx *ctx = nullptr;
    int slot = 0;
    for (int k = first; k < n_frames; k += step, ++slot) {
        const FrameTransform base = transform_of(k);
        gfw_kernel_params params[3]; gfw_buffers planes[3]; int types[3];
        for (int i = 0; i < 3; ++i) {
            FrameTransform t = stab[i].get_frame_transform_at<Luma16>(base, b[i]);
            t.kernel_params.plane_index = i;
            params[i] = t.kernel_params; planes[i] = b[i].to_abi(); types[i] = Luma16::ID;
        }
        if (!ctx) {
            ctx = gfw_create(&params[0], Luma16::ID, GFW_MODEL_OPENCV_FISHEYE, GFW_MODEL_NONE, &planes[0], 0);
            CHECK(ctx != nullptr);
            CHECK(gfw_set_option(ctx, GFW_OPT_SYNCHRONOUS, 0) == GFW_OK);
            CHECK(gfw_set_frame_checksums(ctx, d_sums, (size_t)mine) == GFW_OK);
        }
        for (int i = 0; i < 3; ++i) HIPOK(hipMemsetAsync(d_dst[i], 0, ps[i] * H, (hipStream_t)gfw_get_stream(ctx)));     // (the checksum covers written bytes only; a clean slate keeps the planes comparable too)
        CHECK(gfw_undistort_frame(ctx, 3, planes, params, types, base.matrices[0].data(), H, nullptr, 0) == GFW_OK);
        if (backends && backends->find(gfw_last_backend(ctx)) == std::string::npos) { *backends += gfw_last_backend(ctx); *backends += ' '; }
    }
    MARK("[render first %d: %d frames enqueued]\n", first, mine);
    if (ctx) {
        CHECK(gfw_synchronize(ctx) == GFW_OK);
        MARK("[render first %d: synchronised]\n", first);
        std::vector<unsigned long long> h((size_t)mine);
        HIPOK(hipMemcpy(h.data(), d_sums, sizeof(unsigned long long) * (size_t)mine, hipMemcpyDeviceToHost));
        slot = 0;
        for (int k = first; k < n_frames; k += step, ++slot) (*sums)[k] = h[slot];
        gfw_destroy(ctx);
        MARK("[render first %d: context destroyed]\n", first);
    }
    for (int i = 0; i < 3; ++i) { HIPOK(hipFree(d_src[i])); HIPOK(hipFree(d_dst[i])); }
    HIPOK(hipFree(d_sums));
    MARK("[render first %d: done]\n", first);
    g_stage.fetch_add(1); alarm(120);
}

int main(int argc, char **argv) {
    if (argc > 1 && std::string(argv[1]) == "exit") {
        // leave main() while the context's FIRST background build is still compiling (GFW_OPT_JIT = 1 starts it at the third frame; four frames of 640 x 360 are
        // over in a millisecond): the library's exit hook must join the build with the compiler's own exit handlers still to come (gfw_jit.hip, Rtc)
        std::signal(SIGALRM, on_alarm); alarm(60);
        if (gfw_list_devices(nullptr, 0) <= 0) { std::printf("no HIP device: %s\n", gfw_last_error()); return 3; }
        std::vector<uint8_t> src1[3] = {pattern(W, H, YS, 1), pattern(CW, H, CS, 2), pattern(CW, H, CS, 3)};
        std::vector<unsigned long long> sums(4, 0);
        std::string be;
        render(0, 0, 1, 4, src1, &sums, &be);
        std::printf("exit-during-build: leaving main (backends so far: %s)\n", be.c_str());
        std::fflush(stdout);
        return 0;
    }
    const int T = argc > 1 ? std::atoi(argv[1]) : 4, N = argc > 2 ? std::atoi(argv[2]) : 36;
    std::signal(SIGALRM, on_alarm); alarm(120);
    const int ndev = gfw_list_devices(nullptr, 0);
    if (ndev <= 0) { std::printf("no HIP device: %s\n", gfw_last_error()); return 3; }
    std::vector<uint8_t> src[3] = {pattern(W, H, YS, 1), pattern(CW, H, CS, 2), pattern(CW, H, CS, 3)};
    std::vector<unsigned long long> ref((size_t)N, 0), got((size_t)N, 0);
    std::string ref_backends;
    render(0, 0, 1, N, src, &ref, &ref_backends);                        // one thread, device 0, every frame in order
    std::vector<std::thread> th;
    std::vector<std::string> backends((size_t)T);
    for (int t = 0; t < T; ++t) th.emplace_back(render, t % ndev, t, T, N, src, &got, &backends[(size_t)t]);
    for (auto &x : th) x.join();
    int bad = 0, zero = 0;
    for (int k = 0; k < N; ++k) { bad += ref[(size_t)k] != got[(size_t)k]; zero += ref[(size_t)k] == 0; }
    std::printf("multi-device: %d device(s), %d threads, %d frames; one thread's backends: %s; frames whose checksum differs: %d; zero checksums: %d\n",
                ndev, T, N, ref_backends.c_str(), bad, zero);
    for (int t = 0; t < T; ++t) std::printf("  thread %d on device %d: %s\n", t, t % ndev, backends[(size_t)t].c_str());
    if (bad || zero) { for (int k = 0; k < N; ++k) if (ref[(size_t)k] != got[(size_t)k]) std::printf("  frame %d: %016llx vs %016llx\n", k, ref[(size_t)k], got[(size_t)k]); return 1; }
    std::printf("multi-device ok\n");
    std::fflush(stdout);
    MARK("[main: leaving]\n");
    return 0;
}
