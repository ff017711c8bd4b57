"""The certified first pass of a RADIAL lens model other than the fisheye — GoPro's inverted polynomial (distortion_models/gopro.rs:25-72) — round 6.

Until round 6 a gopro clip evaluated the exact mid-row projection, ten Newton steps and all, for EVERY pixel just to round it to a matrix row (92 us per C2 frame
against the fisheye's 42).  The host now derives a certificate for it (gfw_api_certificate.inc: p1_prepare_radial_gopro — the iteration's contraction, its float
noise floor, a table of the scale over r) and the specialised kernel takes the per-pixel table form of the first pass; pixels within E of a half-integer go to the
exact code as always.  Here: the audit instantiation (a specialised audit build) re-derives the exact row of every certified pixel — none may differ, the measured
|table - exact| must stay inside E — and the frames equal the oracle bit for bit, on the certified path and on the exact one."""
import math

import numpy as np
import pytest

from gyroflow_amd import abi, synthetic as S, warp
import _oracle as O
from test_gpu_parity import assert_plane_equal

pytestmark = pytest.mark.gpu

GOPRO_K = [0.0, 1.0, 0.01, -0.12, 0.02, 0.01, -0.004]


def gopro_lens(w, h, k=GOPRO_K, r_limit=2.5, f_scale=0.47):
    lens = S.gopro_style_lens(w, h)
    lens["model"], lens["k"] = "gopro", list(k) + [0.0] * (12 - len(k))
    lens["f"] = (f_scale * w, f_scale * w)
    if r_limit:
        lens["r_limit"] = r_limit
    return lens


def audit(fr, variant=3):
    outs = [pl["dst"].copy() for pl in fr.planes]
    bufs = [warp.host_buffers(pl["src"], pl["size"], o, pl["out_size"]) for pl, o in zip(fr.planes, outs)]
    params, types = [pl["params"] for pl in fr.planes], [pl["pixel_type"] for pl in fr.planes]
    be = warp.Backend(params[0], types[0], fr.model, fr.digital, bufs[0])
    try:
        be.set_option(abi.OPT_KERNEL_VARIANT, variant)
        be.get_audit(reset=True)
        be.undistort_frame(bufs, params, types, fr.matrices)
        return warp.last_backend(), be.get_audit_full(), outs
    finally:
        be.close()


@pytest.mark.parametrize("fmt", ["YUV422P16LE", "NV12", "YUV420P"])
@pytest.mark.parametrize("fov,r_limit", [(1.0, 2.5), (1.5, 0.0), (0.8, 2.5)])
def test_gopro_certificates_never_disagree_with_the_exact_row(fmt, fov, r_limit):
    w, h = 960, 540
    fr = S.SyntheticFrame(fmt, w, h, seed=0x60 + int(10 * fov), lens=gopro_lens(w, h, r_limit=r_limit), fov=fov, readout_ms=14.0)
    backend, a, outs = audit(fr)
    assert backend == "yuv_fused_p1_jit", backend                       # the audit of a table over r is a specialised build (gfw_api_bake.inc)
    assert a["certified1_wrong"] == 0 and a["out_of_range"] == 0, a
    assert a["certified1"] + a["queued1"] + a["queue_overflow"] == w * h, a
    assert a["pass1_eps_px"] > 0.0 and a["pass1_gap_px"] < a["pass1_eps_px"], a
    assert a["certified1"] > 0.7 * w * h, a                                # the certificate is worth having
    for i, (x, y) in enumerate(zip(O.run_frame(fr), outs)):
        assert_plane_equal(x, y, fr.planes[i]["pixel_type"], "gopro audit build, plane %d" % i)


@pytest.mark.parametrize("fmt", ["YUV422P16LE", "NV12", "P010", "RGBA"])
@pytest.mark.parametrize("interp", [2, 4, 8])
def test_gopro_clips_equal_the_oracle_on_the_certified_and_on_the_exact_pass(fmt, interp):
    w, h = 640, 360
    fr = S.SyntheticFrame(fmt, w, h, seed=0x6F + interp, lens=gopro_lens(w, h), fov=1.2, interpolation=interp)
    ref = O.run_frame(fr)
    got = warp.run_frame(fr, jit=2)
    assert warp.last_backend() == "yuv_fused_p1_jit", warp.last_backend()
    for i, (x, y) in enumerate(zip(ref, got)):
        assert_plane_equal(x, y, fr.planes[i]["pixel_type"], "gopro certified pass, plane %d" % i)
    got = warp.run_frame(fr, jit=0)                                      # ahead of time: the generic-model kernel's exact first pass (a table over r needs a specialised build)
    assert warp.last_backend() == "yuv_fused", warp.last_backend()
    for i, (x, y) in enumerate(zip(ref, got)):
        assert_plane_equal(x, y, fr.planes[i]["pixel_type"], "gopro exact pass, plane %d" % i)
    got = warp.run_frame(fr, variant=2, jit=2)                           # the exact first pass on request (a kernel variant: ahead of time by design, gfw_api_bake.inc)
    assert warp.last_backend() == "yuv_fused", warp.last_backend()
    for i, (x, y) in enumerate(zip(ref, got)):
        assert_plane_equal(x, y, fr.planes[i]["pixel_type"], "gopro exact pass (variant 2), plane %d" % i)


def rot(ax, ay, az):
    cx, sx, cy, sy, cz, sz = math.cos(ax), math.sin(ax), math.cos(ay), math.sin(ay), math.cos(az), math.sin(az)
    rx = np.array([[1, 0, 0], [0, cx, -sx], [0, sx, cx]])
    ry = np.array([[cy, 0, sy], [0, 1, 0], [-sy, 0, cy]])
    rz = np.array([[cz, -sz, 0], [sz, cz, 0], [0, 0, 1]])
    return rz @ ry @ rx


def test_sixty_random_gopro_clips_never_produce_a_wrong_certificate():
    """Random POLY coefficients around a real lens's (the quadratic term included: the reason the table runs over r), focal length, field of view, rotation up to 12
    degrees, readout +-25 ms, both shutter directions: whatever the host certifies the kernel must get right; what it declines (the Newton iteration not provably
    contracting, POLY' too small, E too wide) runs the exact pass."""
    rng = np.random.default_rng(0x60B0)
    served = pixels = certified = 0
    worst = 0.0
    for i in range(60):
        w, h = [(320, 180), (640, 360), (960, 540), (1280, 720)][int(rng.integers(0, 4))]
        k = [0.0, rng.uniform(0.9, 1.1), rng.uniform(-0.03, 0.03), rng.uniform(-0.2, 0.02), rng.uniform(-0.04, 0.04), rng.uniform(-0.02, 0.02), rng.uniform(-0.008, 0.008)]
        lens = gopro_lens(w, h, k=k, r_limit=float(rng.choice([0.0, 2.0, 3.0])), f_scale=rng.uniform(0.4, 0.9))
        hrs = bool(rng.integers(0, 4) == 0)
        readout = rng.uniform(-25.0, 25.0)
        fr = S.SyntheticFrame("YUV422P16LE" if rng.integers(0, 2) else "NV12", w, h, seed=int(rng.integers(1, 1 << 30)), lens=lens, fov=rng.uniform(0.6, 2.0),
                              readout_ms=readout if abs(readout) > 0.5 else 8.0, horizontal_rs=hrs)
        rows = fr.matrices.shape[0]
        base = np.radians(rng.uniform(-12.0, 12.0, 3))
        rate = np.radians(rng.uniform(-200.0, 200.0, 3)) * (readout / 1000.0)
        nk = S.new_k(lens, fr.planes[0]["params"].fov, w, h)
        t = (np.arange(rows) / max(rows - 1, 1)) - 0.5
        m = np.zeros((rows, 14), dtype=np.float32)
        for y in range(rows):
            r = rot(*(base + rate * t[y]))
            r[0, 1] *= -1.0; r[0, 2] *= -1.0; r[1, 0] *= -1.0; r[2, 0] *= -1.0
            m[y, :9] = np.linalg.inv(nk @ r).reshape(9).astype(np.float32)
        fr.matrices = m
        backend, a, outs = audit(fr)
        for j, (x, y) in enumerate(zip(O.run_frame(fr), outs)):
            assert_plane_equal(x, y, fr.planes[j]["pixel_type"], "random gopro clip %d (%s), plane %d" % (i, backend, j))
        if backend != "yuv_fused_p1_jit":
            continue
        served += 1
        assert a["certified1_wrong"] == 0 and a["out_of_range"] == 0, (i, k, a)
        assert a["certified1"] + a["queued1"] + a["queue_overflow"] == w * h, (i, a)
        assert a["pass1_gap_px"] < a["pass1_eps_px"], (i, k, a)
        worst = max(worst, a["pass1_gap_px"] / a["pass1_eps_px"])
        pixels += w * h
        certified += a["certified1"]
    print("gopro certified first pass: %d of 60 clips served, %.1f %% of their pixels certified, 0 wrong; worst gap / E = %.3f" % (served, 100.0 * certified / max(pixels, 1), worst))
    assert served >= 30, served


# ---- the closed-form radial models whose clips take the certified pass: Sony / generic polynomial (polynomials in theta = atan r; sony.rs:69-88, generic_polynomial.rs) -----
# (lensfun's poly3 / poly5 / ptlens have a certificate too — tests/test_emu_pass1_audit.py audits it in the interpreter — but their exact projection is cheaper than
# the table: measured 47 -> 50 us on C2, profiles/r06_radial_closed_form.txt; their clips keep the exact first pass)
CLOSED_FORM_K = {
    "sony": [1.0, 0.01, -0.05, 0.02, 0.003, -0.001],
    "generic_polynomial": [1.0, 0.01, -0.05, 0.02, 0.003, -0.001, 0.0005, -0.0002, 0.0001, 0.0, 0.0, 0.00002],
}
CLOSED_FORM_RANGES = {
    "sony": [(0.9, 1.1), (-0.05, 0.05), (-0.15, 0.05), (-0.04, 0.04), (-0.01, 0.01), (-0.004, 0.004)],
    "generic_polynomial": [(0.9, 1.1), (-0.05, 0.05), (-0.15, 0.05), (-0.04, 0.04), (-0.01, 0.01), (-0.004, 0.004)] + [(-0.001, 0.001)] * 6,
}


def closed_form_lens(model, w, h, k=None, r_limit=2.5, f_scale=0.47):
    lens = gopro_lens(w, h, k=k if k is not None else CLOSED_FORM_K[model], r_limit=r_limit, f_scale=f_scale)
    lens["model"] = model
    return lens


@pytest.mark.parametrize("fmt", ["YUV422P16LE", "NV12"])
@pytest.mark.parametrize("model", sorted(CLOSED_FORM_K))
@pytest.mark.parametrize("fov,r_limit", [(1.0, 2.5), (1.5, 0.0)])
def test_closed_form_certificates_never_disagree_with_the_exact_row(model, fmt, fov, r_limit):
    w, h = 960, 540
    fr = S.SyntheticFrame(fmt, w, h, seed=0x70 + int(10 * fov), lens=closed_form_lens(model, w, h, r_limit=r_limit), fov=fov, readout_ms=14.0)
    backend, a, outs = audit(fr)
    assert backend == "yuv_fused_p1_jit", backend
    assert a["certified1_wrong"] == 0 and a["out_of_range"] == 0, a
    assert a["certified1"] + a["queued1"] + a["queue_overflow"] == w * h, a
    assert a["pass1_eps_px"] > 0.0 and a["pass1_gap_px"] < a["pass1_eps_px"], a
    assert a["certified1"] > 0.8 * w * h, a
    for i, (x, y) in enumerate(zip(O.run_frame(fr), outs)):
        assert_plane_equal(x, y, fr.planes[i]["pixel_type"], "%s audit build, plane %d" % (model, i))


@pytest.mark.parametrize("model", sorted(CLOSED_FORM_K))
@pytest.mark.parametrize("fmt,interp", [("YUV422P16LE", 2), ("NV12", 4), ("P010", 8), ("RGBA", 2)])
def test_closed_form_clips_equal_the_oracle_on_the_certified_and_on_the_exact_pass(model, fmt, interp):
    w, h = 640, 360
    fr = S.SyntheticFrame(fmt, w, h, seed=0x7F + interp, lens=closed_form_lens(model, w, h), fov=1.2, interpolation=interp)
    ref = O.run_frame(fr)
    got = warp.run_frame(fr, jit=2)
    assert warp.last_backend() == "yuv_fused_p1_jit", warp.last_backend()
    for i, (x, y) in enumerate(zip(ref, got)):
        assert_plane_equal(x, y, fr.planes[i]["pixel_type"], "%s certified pass, plane %d" % (model, i))
    got = warp.run_frame(fr, jit=0)
    assert warp.last_backend() == "yuv_fused", warp.last_backend()
    for i, (x, y) in enumerate(zip(ref, got)):
        assert_plane_equal(x, y, fr.planes[i]["pixel_type"], "%s exact pass, plane %d" % (model, i))


@pytest.mark.parametrize("model", sorted(CLOSED_FORM_K))
def test_twenty_random_closed_form_clips_never_produce_a_wrong_certificate(model):
    rng = np.random.default_rng(0x70B0 + len(model) * 7)
    served = pixels = certified = 0
    worst = 0.0
    for i in range(20):
        w, h = [(320, 180), (640, 360), (960, 540), (1280, 720)][int(rng.integers(0, 4))]
        k = [float(rng.uniform(lo, hi)) for lo, hi in CLOSED_FORM_RANGES[model]]
        lens = closed_form_lens(model, w, h, k=k, r_limit=float(rng.choice([0.0, 2.0, 3.0])), f_scale=rng.uniform(0.4, 0.9))
        hrs = bool(rng.integers(0, 4) == 0)
        readout = rng.uniform(-25.0, 25.0)
        fr = S.SyntheticFrame("YUV422P16LE" if rng.integers(0, 2) else "NV12", w, h, seed=int(rng.integers(1, 1 << 30)), lens=lens, fov=rng.uniform(0.6, 2.0),
                              readout_ms=readout if abs(readout) > 0.5 else 8.0, horizontal_rs=hrs)
        rows = fr.matrices.shape[0]
        base = np.radians(rng.uniform(-12.0, 12.0, 3))
        rate = np.radians(rng.uniform(-200.0, 200.0, 3)) * (readout / 1000.0)
        nk = S.new_k(lens, fr.planes[0]["params"].fov, w, h)
        t = (np.arange(rows) / max(rows - 1, 1)) - 0.5
        m = np.zeros((rows, 14), dtype=np.float32)
        for y in range(rows):
            r = rot(*(base + rate * t[y]))
            r[0, 1] *= -1.0; r[0, 2] *= -1.0; r[1, 0] *= -1.0; r[2, 0] *= -1.0
            m[y, :9] = np.linalg.inv(nk @ r).reshape(9).astype(np.float32)
        fr.matrices = m
        backend, a, outs = audit(fr)
        for j, (x, y) in enumerate(zip(O.run_frame(fr), outs)):
            assert_plane_equal(x, y, fr.planes[j]["pixel_type"], "random %s clip %d (%s), plane %d" % (model, i, backend, j))
        if backend != "yuv_fused_p1_jit":
            continue
        served += 1
        assert a["certified1_wrong"] == 0 and a["out_of_range"] == 0, (i, k, a)
        assert a["certified1"] + a["queued1"] + a["queue_overflow"] == w * h, (i, a)
        assert a["pass1_gap_px"] < a["pass1_eps_px"], (i, k, a)
        worst = max(worst, a["pass1_gap_px"] / a["pass1_eps_px"])
        pixels += w * h
        certified += a["certified1"]
    print("%s certified first pass: %d of 20 clips served, %.1f %% of their pixels certified, 0 wrong; worst gap / E = %.3f" % (model, served, 100.0 * certified / max(pixels, 1), worst))
    assert served >= 15, served


@pytest.mark.parametrize("model", ["poly3", "poly5", "ptlens"])
def test_polynomials_in_r_keep_the_exact_first_pass(model):
    w, h = 640, 360
    k = {"poly3": [0.06], "poly5": [0.08, -0.02], "ptlens": [0.01, -0.03, 0.02]}[model]
    fr = S.SyntheticFrame("YUV422P16LE", w, h, seed=0x81, lens=closed_form_lens(model, w, h, k=k), fov=1.2)
    got = warp.run_frame(fr, jit=2)
    assert warp.last_backend() == "yuv_fused_jit", warp.last_backend()
    for i, (x, y) in enumerate(zip(O.run_frame(fr), got)):
        assert_plane_equal(x, y, fr.planes[i]["pixel_type"], "%s, plane %d" % (model, i))


@pytest.mark.parametrize("model", ["gopro", "sony", "generic_polynomial"])
@pytest.mark.parametrize("fmt,n", [("YUV422P16LE", 5), ("NV12", 3)])
def test_every_frame_of_a_clip_launch_is_written_on_the_certified_pass(model, fmt, n):
    """Several frames per launch (gfw_undistort_clip, device-resident tables) through a radial model's certified build.  Round 6: the generic polynomial's first such
    build kept its argument block in scratch — a pointer test in the kernel body stopped the optimiser from reading the argument segment — and, the block sitting
    at private offset 0 where that very test read "null", took the launch for a single frame: frames 1.. of every clip call were never written (caught by bench.py's
    oracle comparison; profiles/r06_radial_closed_form.txt).  Every frame must land in its own planes, equal to the oracle's."""
    import test_gpu_jit as J
    w, h = 640, 360
    lens = gopro_lens(w, h) if model == "gopro" else closed_form_lens(model, w, h, r_limit=0.0)
    frames = [S.SyntheticFrame(fmt, w, h, seed=0x9F10 + j, timestamp_ms=1000.0 + 33.3 * j, lens=dict(lens), readout_ms=16.0, pixels=False) for j in range(n)]
    backend, status, (ms, launches, covered), outs, srcs = J.device_clip(frames, 2, True)
    assert backend == "yuv_fused_p1_jit" and status[0] == 2 and launches == 1 and covered == n, (backend, status, launches, covered)
    for j, fr in enumerate(frames):
        for p, (a, b) in enumerate(zip(O.run_frame(J._View(fr, srcs[j])), outs[j])):
            assert_plane_equal(a, b, fr.planes[p]["pixel_type"], "%s clip launch, frame %d plane %d" % (model, j, p))
