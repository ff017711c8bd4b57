"""The run-time specialised kernel (gfw_jit.hip: the fused kernel compiled per clip by hiprtc with the clip's constants baked in, the
way the reference compiles its OpenCL source per clip, opencl.rs:181-214) and the clip entry point built on it: same bits as the
oracle and as the ahead-of-time kernels, through every template family it can be asked for; the asynchronous hand-over; the
multi-frame launches of gfw_undistort_clip."""
import time

import numpy as np
import pytest

from gyroflow_amd import abi, synthetic as S, warp
import _oracle as O
from test_gpu_parity import assert_plane_equal
from test_gpu_fullsize import _View

pytestmark = pytest.mark.gpu


def check_jit(fr, what):
    ref = O.run_frame(fr)
    got = warp.run_frame(fr, jit=2)
    assert warp.last_backend().endswith("_jit"), (what, warp.last_backend())
    for i, (a, b) in enumerate(zip(ref, got)):
        assert_plane_equal(a, b, fr.planes[i]["pixel_type"], "%s: specialised kernel, plane %d" % (what, i))
    aot = warp.run_frame(fr, jit=0)
    assert not warp.last_backend().endswith("_jit")
    for i, (a, b) in enumerate(zip(ref, aot)):
        assert_plane_equal(a, b, fr.planes[i]["pixel_type"], "%s: ahead-of-time kernel, plane %d" % (what, i))


@pytest.mark.parametrize("fmt", ["NV12", "P010", "P210", "YUV420P", "YUV420P10LE", "YUV422P16LE", "YUV444P16LE", "GBRAPF32LE", "RGBA", "RGBA64", "RGBAF32"])
def test_formats(fmt):
    check_jit(S.SyntheticFrame(fmt, 320, 192, seed=11), fmt)


@pytest.mark.parametrize("interp", [2, 4, 8])
@pytest.mark.parametrize("bg", [0, 1, 2])
def test_samplers_and_background_modes(interp, bg):
    fr = S.SyntheticFrame("YUV422P16LE", 322, 190, seed=40 + interp + bg, fov=2.2, interpolation=interp, base_overrides={"background_mode": bg},
                          background_rgba=(0.25, 0.5, 0.75, 1.0))
    check_jit(fr, "interpolation %d background mode %d" % (interp, bg))


# (Lanczos4 over f32 copies of the planes — one HBM pass instead of the conversions in the tap loop — was built, passed parity and measured
# SLOWER: 206 + 25 us against 182 us per C2 frame, the tap-row fetches are the bound, not the conversions: profiles/r03_lanczos4_f32_source.txt)


@pytest.mark.parametrize("digital", ["gopro_superview", "gopro6_superview", "gopro_hyperview", "gopro_warp", "digital_stretch"])
def test_digital_lens_on_top_of_the_fisheye(digital):
    """GoPro SuperView / HyperView clips: ahead of time they run the generic-model instantiation; baked, the digital lens is a literal and the
    specialised fisheye projection serves them (exact first pass)."""
    from test_gpu_lens_models import DIGITAL
    lens = S.gopro_style_lens(320, 192)
    lens["digital"] = digital
    fr = S.SyntheticFrame("YUV422P16LE", 320, 192, seed=71, lens=lens, fov=1.3, base_overrides={"digital_lens_params": DIGITAL[digital]})
    assert fr.planes[0]["params"].flags & abi.FLAG_HAS_DIGITAL_LENS
    check_jit(fr, digital)
    assert warp.last_backend() == "yuv_fused"                 # the ahead-of-time run that check_jit ends with


@pytest.mark.parametrize("model", ["opencv_standard", "poly3", "poly5", "ptlens", "insta360", "sony", "generic_polynomial", "gopro", "opencv_fisheye"])
@pytest.mark.parametrize("lca", [1.0, 0.45])
def test_every_lens_model_specialised(model, lca):
    """Every physical lens model — and the fisheye under the lens-correction blend — through the generic-model body with the model as a literal
    (GFW_JIT_MODEL = -1: the switch over the lens models folds to the one in use)."""
    from test_gpu_lens_models import PHYSICAL
    if model == "opencv_fisheye" and lca == 1.0:
        pytest.skip("the lean fisheye instantiation: every other test of this file")
    w, h = 256, 160
    lens = S.gopro_style_lens(w, h)
    lens["model"] = model
    lens["k"] = PHYSICAL[model] + [0.0] * (12 - len(PHYSICAL[model]))
    if model == "gopro":
        lens["r_limit"] = 2.5
    check_jit(S.SyntheticFrame("YUV422P16LE", w, h, seed=31, lens=lens, fov=1.2, base_overrides={"lens_correction_amount": lca}), "%s lca %g" % (model, lca))


@pytest.mark.parametrize("interp", [2, 8])
def test_generic_features_specialised(interp):
    """Refraction, blend under a digital lens, IBIS/OIS terms, background mode 3 (MODEL = -2): feature bits as literals."""
    from test_gpu_lens_models import DIGITAL, PHYSICAL
    w, h = 256, 160
    check_jit(S.SyntheticFrame("YUV422P16LE", w, h, seed=61, fov=1.3, interpolation=interp, base_overrides={"light_refraction_coefficient": 1.33}), "refraction")
    lens = S.gopro_style_lens(w, h)
    lens["digital"] = "gopro_superview"
    check_jit(S.SyntheticFrame("NV12", w, h, seed=62, lens=lens, fov=1.1, interpolation=interp,
                               base_overrides={"lens_correction_amount": 0.6, "digital_lens_params": DIGITAL["gopro_superview"]}), "digital lens + blend")
    lens = S.gopro_style_lens(w, h)
    lens["model"] = "poly5"
    lens["k"] = PHYSICAL["poly5"] + [0.0] * (12 - len(PHYSICAL["poly5"]))
    lens["digital"] = "digital_stretch"
    check_jit(S.SyntheticFrame("YUV420P", w, h, seed=63, lens=lens, fov=1.1, interpolation=interp,
                               base_overrides={"digital_lens_params": DIGITAL["digital_stretch"]}), "poly5 under a digital lens")
    fr = S.SyntheticFrame("YUV422P16LE", w, h, seed=53, fov=1.2, interpolation=interp)
    y = np.arange(fr.matrices.shape[0], dtype=np.float32)
    fr.matrices[:, 9] = 1.5 * np.sin(y * 0.05)
    fr.matrices[:, 10] = -0.8 * np.cos(y * 0.03)
    fr.matrices[:, 11] = 0.004 * np.sin(y * 0.02)
    fr.matrices[:, 12] = 0.6
    fr.matrices[:, 13] = -0.4
    check_jit(fr, "IBIS / OIS terms")
    fr = S.SyntheticFrame("YUV422P16LE", 322, 190, seed=64, fov=2.2, interpolation=interp,
                          base_overrides={"background_mode": 3, "background_margin": 0.08, "background_margin_feather": 0.12})
    check_jit(fr, "background mode 3")


@pytest.mark.parametrize("with_mesh,with_fpd", [(True, True), (False, True)])
def test_sony_mesh_specialised(with_mesh, with_fpd):
    from test_gpu_lens_models import synthetic_mesh
    w, h = 192, 128
    fr = S.SyntheticFrame("NV12", w, h, seed=47, fov=1.1)
    mesh = np.asarray(synthetic_mesh(w, h, with_fpd, with_mesh), dtype=np.float32)
    ref = []
    for pl in fr.planes:
        dst = pl["dst"].copy()
        assert O.undistort_image(pl["src"], pl["size"], dst, pl["out_size"], pl["params"], pl["pixel_type"], fr.model, fr.digital, fr.matrices, mesh=mesh) == 1
        ref.append(dst)
    params = [pl["params"] for pl in fr.planes]
    types = [pl["pixel_type"] for pl in fr.planes]
    for jit in (2, 0):
        outs = [pl["dst"].copy() for pl in fr.planes]
        bufs = [warp.host_buffers(pl["src"], pl["size"], o, pl["out_size"]) for pl, o in zip(fr.planes, outs)]
        be = warp.Backend(params[0], types[0], fr.model, fr.digital, bufs[0])
        try:
            be.set_option(abi.OPT_JIT, jit)
            be.undistort_frame(bufs, params, types, fr.matrices, mesh=mesh)
            assert warp.last_backend() == ("yuv_fused_jit" if jit else "yuv_fused"), warp.last_backend()
        finally:
            be.close()
        for i, (a, b) in enumerate(zip(ref, outs)):
            assert_plane_equal(a, b, fr.planes[i]["pixel_type"], "mesh, jit %d, plane %d" % (jit, i))


def test_geometry_variants():
    check_jit(S.SyntheticFrame("YUV422P16LE", 256, 160, seed=23, readout_ms=0.0), "one matrix")
    check_jit(S.SyntheticFrame("YUV422P16LE", 256, 160, seed=23, horizontal_rs=True), "horizontal rolling shutter")
    check_jit(S.SyntheticFrame("NV12", 130, 66, seed=19, fov=3.0), "odd size, zoomed out")
    lens = S.gopro_style_lens(320, 192)
    lens["r_limit"] = 0.6
    check_jit(S.SyntheticFrame("YUV422P16LE", 320, 192, seed=29, lens=lens, fov=1.5), "r_limit")
    lens = S.gopro_style_lens(320, 192)
    lens["k"] = [0.0] * 12
    check_jit(S.SyntheticFrame("YUV422P16LE", 320, 192, seed=31, lens=lens, readout_ms=20.0), "all-zero k")
    check_jit(S.SyntheticFrame("RGBAF32", 320, 192, seed=33, fov=0.82, base_overrides={"translation2d": (13.25, -7.5)}), "adaptive-zoom crop")


def device_clip(frames, jit_mode, use_clip, reps=1, shared_dst=False, chain=False):
    """Frames through HIP_DEVICE buffers and device-resident tables on one context: frame by frame, or gfw_undistort_clip.
    Returns (backend of the last call, jit status, outputs per frame)."""
    import torch
    dev = torch.device("cuda", 0)
    d_src = [fr.device_planes(dev) for fr in frames]
    d_dst = [fr.device_outputs(dev) for fr in frames]
    d_mat = [torch.from_numpy(warp.pack_matrices(fr.matrices)).to(dev) for fr in frames]
    torch.cuda.synchronize(dev)
    types = [pl["pixel_type"] for pl in frames[0].planes]
    params = [pl["params"] for pl in frames[0].planes]
    if shared_dst:                                    # every frame writes frame 0's destination
        d_dst = [d_dst[0]] * len(frames)
    if chain:                                         # frame j reads what frame j - 1 wrote (same plane sizes on both sides)
        d_src = [d_src[0]] + d_dst[:-1]
    bufs = [[warp.device_buffers(d_src[j][p].data_ptr(), d_src[j][p].numel(), pl["size"], d_dst[j][p].data_ptr(), d_dst[j][p].numel(), pl["out_size"])
             for p, pl in enumerate(fr.planes)] for j, fr in enumerate(frames)]
    rows = frames[0].matrices.shape[0]
    be = warp.Backend(params[0], types[0], frames[0].model, frames[0].digital, bufs[0][0])
    try:
        be.set_stream(torch.cuda.current_stream(dev).cuda_stream)
        be.set_option(abi.OPT_SYNCHRONOUS, 0)
        be.set_option(abi.OPT_MATRICES_ON_DEVICE, 2)
        be.set_option(abi.OPT_JIT, jit_mode)
        be.set_option(abi.OPT_PROFILE, 1)
        for _ in range(reps):
            if use_clip:
                warp.ClipCall(be, bufs, params, types, [m.data_ptr() for m in d_mat], rows)()
            else:
                for j in range(len(frames)):
                    warp.FrameCall(be, bufs[j], params, types, d_mat[j].data_ptr(), rows)()
        be.synchronize()
        backend, status = warp.last_backend(), be.jit_status()
        prof = be.get_profile_frames()
    finally:
        be.close()
    torch.cuda.synchronize(dev)
    return backend, status, prof, [[t.cpu().numpy() for t in d_dst[j]] for j in range(len(frames))], [[t.cpu().numpy() for t in d_src[j]] for j in range(len(frames))]


@pytest.mark.parametrize("fmt,n", [("YUV422P16LE", 19), ("NV12", 8), ("RGBA64", 3)])
def test_clip_entry_point_matches_frame_by_frame_and_the_oracle(fmt, n):
    frames = [S.SyntheticFrame(fmt, 320, 192, seed=0x9F10 + j, timestamp_ms=1000.0 + 33.3 * j, pixels=False) for j in range(n)]
    backend, status, (ms, launches, covered), outs, srcs = device_clip(frames, 2, True)
    assert backend.endswith("_jit") and status[0] == 2, (backend, status)
    assert covered == n and launches == (n + abi.CLIP_MAX - 1) // abi.CLIP_MAX, (launches, covered)          # launches of up to GFW_CLIP_FRAMES_MAX frames
    _, _, _, outs_fb, _ = device_clip(frames, 0, False)
    for j, fr in enumerate(frames):
        ref = O.run_frame(_View(fr, srcs[j]))
        for p, (a, b, c) in enumerate(zip(ref, outs[j], outs_fb[j])):
            assert_plane_equal(a, b, fr.planes[p]["pixel_type"], "clip launch, frame %d plane %d" % (j, p))
            assert_plane_equal(a, c, fr.planes[p]["pixel_type"], "frame by frame, frame %d plane %d" % (j, p))


_CAP_SCRIPT = r"""
import sys, json
sys.path.insert(0, %(root)r); sys.path.insert(0, %(tests)r)
import numpy as np
from gyroflow_amd import synthetic as S
import _oracle as O
import test_gpu_jit as J
frames = [S.SyntheticFrame("YUV422P16LE", 320, 192, seed=0x9F10 + j, timestamp_ms=1000.0 + 33.3 * j, pixels=False) for j in range(7)]
backend, status, (ms, launches, covered), outs, srcs = J.device_clip(frames, 2, True)
bad = 0
for j, fr in enumerate(frames):
    for a, b in zip(O.run_frame(J._View(fr, srcs[j])), outs[j]):
        bad += int(np.count_nonzero(np.asarray(a) != np.asarray(b)))
print("RESULT " + json.dumps({"backend": backend, "launches": int(launches), "covered": int(covered), "bad": bad}))
"""


def test_a_clip_call_is_dealt_evenly_over_launches_of_capped_size(tmp_path):
    """gfw_api.hip clip_launch_limit: a launch of the specialised kernel takes at most GFW_CLIP_LAUNCH_MB of source + destination (1.1 GB by default: 8K frames
    in launches of sixteen measured 9 % slower than in launches of four, profiles/r06_c3_frames_per_launch.txt) and the call's frames are dealt evenly over the
    launches that needs.  With the cap at 1 MB a 320 x 192 4:2:2 16-bit frame (491 KB) goes two to a launch: a 7-frame call leaves as 2 + 2 + 2 + 1, every frame
    bit-exact.  (The cap is read once per process: its own interpreter.)"""
    import json, os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = tmp_path / "cap.py"
    script.write_text(_CAP_SCRIPT % {"root": root, "tests": os.path.join(root, "tests")})
    for mb, want in (("1", 4), ("0", 1)):                 # 0: the default cap — one launch of seven
        r = subprocess.run([sys.executable, str(script)], env=dict(os.environ, GFW_CLIP_LAUNCH_MB=mb), capture_output=True, text=True, timeout=600)
        line = [l for l in r.stdout.splitlines() if l.startswith("RESULT ")]
        assert r.returncode == 0 and line, (r.stdout[-2000:], r.stderr[-2000:])
        res = json.loads(line[-1][7:])
        assert res["backend"].endswith("_jit") and res["launches"] == want and res["covered"] == 7 and res["bad"] == 0, (mb, res)


@pytest.mark.parametrize("fmt,kw", [
    ("YUV422P16LE", dict(fov=1.4, base_overrides={"background_mode": 3, "background_margin": 0.1, "background_margin_feather": 0.1})),
    ("YUVA444P10LE", dict(fov=1.3, interpolation=4, base_overrides={"background_mode": 3, "background_margin": 0.13, "background_margin_feather": 0.24})),
    ("YUV420P", dict(fov=1.2, base_overrides={"lens_correction_amount": 0.5})),
    ("P010", dict(fov=1.5, base_overrides={"background_mode": 3, "background_margin": 0.05, "background_margin_feather": 0.05, "light_refraction_coefficient": 0.9}, flags=2048)),
])
def test_clip_launches_of_the_generic_model_bodies(fmt, kw):
    """Several frames per launch through the feature-bit bodies (MODEL = -1 / -2).  Background mode 3 on planar chroma used to take the chroma
    planes' pointers from the argument block — frame 0's — for every frame of the launch (found by the host interpreter's random sweep,
    tests/test_emu_kernel.py::test_clip_launch_of_the_feature_bodies): every frame must land in its own planes."""
    frames = [S.SyntheticFrame(fmt, 192, 112, seed=100 + j, timestamp_ms=1000.0 + 33.3 * j, pixels=False, **kw) for j in range(5)]
    backend, status, (ms, launches, covered), outs, srcs = device_clip(frames, 2, True)
    assert backend.endswith("_jit") and status[0] == 2 and launches == 1 and covered == 5, (backend, status, launches, covered)
    for j, fr in enumerate(frames):
        ref = O.run_frame(_View(fr, srcs[j]))
        for p, (a, b) in enumerate(zip(ref, outs[j])):
            assert_plane_equal(a, b, fr.planes[p]["pixel_type"], "clip launch, frame %d plane %d" % (j, p))


def test_clip_frames_that_touch_a_pending_frames_buffers_leave_in_order():
    """The calls a clip stands for are ordered.  Frames that all write one destination go out one by one (the last one wins); a frame that
    reads the previous frame's output waits for it — same bytes as the frame-by-frame loop either way."""
    frames = [S.SyntheticFrame("YUV422P16LE", 320, 192, seed=0x9F10 + j, timestamp_ms=1000.0 + 33.3 * j, pixels=False) for j in range(5)]
    backend, status, (ms, launches, covered), outs, srcs = device_clip(frames, 2, True, shared_dst=True)
    assert backend.endswith("_jit") and covered == 5 and launches == 5, (backend, launches, covered)
    ref = O.run_frame(_View(frames[4], srcs[4]))
    for p, (a, b) in enumerate(zip(ref, outs[0])):
        assert_plane_equal(a, b, frames[4].planes[p]["pixel_type"], "shared destination, plane %d" % p)
    frames = [S.SyntheticFrame("YUV444P16LE", 320, 192, seed=0x9F10 + j, timestamp_ms=1000.0 + 33.3 * j, pixels=False) for j in range(4)]
    backend, status, (ms, launches, covered), outs, srcs = device_clip(frames, 2, True, chain=True)
    assert backend.endswith("_jit") and covered == 4 and launches == 4, (backend, launches, covered)
    _, _, _, outs_fb, _ = device_clip(frames, 0, False, chain=True)
    for j in range(4):
        for p, (a, b) in enumerate(zip(outs_fb[j], outs[j])):
            assert np.array_equal(a, b), "chained frames, frame %d plane %d" % (j, p)


def test_clip_entry_point_without_the_specialised_kernel_runs_frame_by_frame():
    frames = [S.SyntheticFrame("YUV422P16LE", 320, 192, seed=0x9F10 + j, timestamp_ms=1000.0 + 33.3 * j, pixels=False) for j in range(5)]
    backend, status, (ms, launches, covered), outs, srcs = device_clip(frames, 0, True)
    assert backend == "yuv_fused_p1" and launches == 5 and covered == 5
    for j, fr in enumerate(frames):
        ref = O.run_frame(_View(fr, srcs[j]))
        for p, (a, b) in enumerate(zip(ref, outs[j])):
            assert_plane_equal(a, b, fr.planes[p]["pixel_type"], "frame %d plane %d" % (j, p))


def test_background_build_takes_over_mid_clip():
    """GFW_OPT_JIT = 1 (the default): the first frames run ahead-of-time while the specialised kernel is built on a worker thread;
    once it is ready the same context switches to it.  Every frame, before and after, is the oracle's."""
    import torch
    frames = [S.SyntheticFrame("YUV422P16LE", 384, 216, seed=0x9F10 + 100 + j, timestamp_ms=500.0 + 33.3 * j, pixels=False) for j in range(4)]
    backend, status, _, outs, srcs = device_clip(frames, 1, False)
    assert backend in ("yuv_fused_p1", "yuv_fused_p1_jit")
    # the build was started by the third frame of the clip; wait for the cache to hold it, then run the clip again
    deadline = time.time() + 60.0
    while time.time() < deadline:
        backend2, status2, _, outs2, _ = device_clip(frames, 1, False)
        if backend2.endswith("_jit"):
            break
        time.sleep(0.5)
    assert backend2 == "yuv_fused_p1_jit" and status2[0] == 2, (backend2, status2)
    for j, fr in enumerate(frames):
        ref = O.run_frame(_View(fr, srcs[j]))
        for p, (a, b, c) in enumerate(zip(ref, outs[j], outs2[j])):
            assert_plane_equal(a, b, fr.planes[p]["pixel_type"], "before the hand-over, frame %d plane %d" % (j, p))
            assert_plane_equal(a, c, fr.planes[p]["pixel_type"], "after the hand-over, frame %d plane %d" % (j, p))


def test_full_size_c2_clip():
    frames = [S.SyntheticFrame("YUV422P16LE", 3840, 2160, seed=0x9F10 + j, timestamp_ms=1000.0 + 33.3 * j, pixels=False) for j in (0, 17, 63)]
    backend, status, (ms, launches, covered), outs, srcs = device_clip(frames, 2, True, reps=2)
    assert backend == "yuv_fused_p1_jit" and covered == 6 and launches == 2
    print("C2 clip launch of 3 frames: %.1f us per frame (compile %.0f ms)" % (1e3 * ms / covered, status[1]))
    for j, fr in enumerate(frames):
        ref = O.run_frame(_View(fr, srcs[j]))
        for p, (a, b) in enumerate(zip(ref, outs[j])):
            assert_plane_equal(a, b, fr.planes[p]["pixel_type"], "frame %d plane %d" % (j, p))


def test_clip_call_under_the_default_synchronous_option_returns_finished_frames():
    """GFW_OPT_SYNCHRONOUS defaults to 1: "the outputs are complete when the call returns".  A frame that joins a clip launch leaves run_planes before its
    own synchronisation point, so gfw_undistort_clip has to synchronise after its last launch (round-3 advisor finding): the outputs are read here through a
    NON-blocking side stream right after the call, with no synchronize of ours in between."""
    import torch
    dev = torch.device("cuda", 0)
    assert abi.load_library().gfw_set_device(0) == 0
    frames = [S.SyntheticFrame("YUV422P16LE", 1920, 1080, seed=0xC11 + j, timestamp_ms=1000.0 + 33.3 * j) for j in range(8)]
    d_src = [fr.device_planes(dev) for fr in frames]
    d_dst = [fr.device_outputs(dev) for fr in frames]
    d_mat = [torch.from_numpy(warp.pack_matrices(fr.matrices)).to(dev) for fr in frames]
    torch.cuda.synchronize(dev)
    types = [pl["pixel_type"] for pl in frames[0].planes]
    params = [pl["params"] for pl in frames[0].planes]
    bufs = [[warp.device_buffers(d_src[j][p].data_ptr(), d_src[j][p].numel(), pl["size"], d_dst[j][p].data_ptr(), d_dst[j][p].numel(), pl["out_size"])
             for p, pl in enumerate(fr.planes)] for j, fr in enumerate(frames)]
    be = warp.Backend(params[0], types[0], frames[0].model, frames[0].digital, bufs[0][0])
    side = torch.cuda.Stream(dev)
    try:
        be.set_option(abi.OPT_MATRICES_ON_DEVICE, 2)                 # (GFW_OPT_SYNCHRONOUS stays at its default)
        be.set_option(abi.OPT_JIT, 2)
        call = warp.ClipCall(be, bufs, params, types, [m.data_ptr() for m in d_mat], frames[0].matrices.shape[0])
        call()                                                       # first call: compiles, frames may run one by one
        for t in d_dst:
            for x in t:
                x.fill_(0x5A)
        torch.cuda.synchronize(dev)
        call()                                                       # now the eight frames share one launch of the specialised kernel
        assert warp.Backend.last_backend_of(be).endswith("_jit"), warp.Backend.last_backend_of(be)
        with torch.cuda.stream(side):                                # the library's own stream is unknown to torch: nothing here waits for it
            host = [[torch.empty_like(x, device="cpu").pin_memory() for x in t] for t in d_dst]
            for hj, tj in zip(host, d_dst):
                for h, x in zip(hj, tj):
                    h.copy_(x, non_blocking=True)
        side.synchronize()
    finally:
        be.close()
    for j, fr in enumerate(frames):
        ref = O.run_frame(_View(fr, [t.cpu().numpy() for t in d_src[j]]))
        for p, (a, b) in enumerate(zip(ref, host[j])):
            assert np.array_equal(a, b.numpy()), "frame %d plane %d was read before the clip call had finished it" % (j, p)
