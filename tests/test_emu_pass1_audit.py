"""The first pass's certificate, audited on the CPU tier: the fused kernel's AUDIT instantiation (ahead-of-time form, host-interpreted: tests/_emu.py) re-derives
the exact rolling-shutter row of every pixel whose approximate row it accepted and records the largest |approximate - exact| coordinate, and range-checks every
tap, store, matrix row and table entry against the declared buffer lengths — over seeded random fisheye clips (lens coefficients, focal length, principal point,
field of view 0.5-3, readout up to +-30 ms, both shutter directions, odd sizes).  The CPU twin of tests/test_gpu_pass1_sweep.py (same assertions: no wrong
certificate, gap below half the certificate's half-width E, nothing out of range, the queue never overflows)."""
import math

import numpy as np
import pytest

from gyroflow_amd import synthetic as S
import _emu
import _oracle as O


def random_clip(seed):
    rng = np.random.default_rng(50000 + seed)
    w, h = int(rng.integers(60, 330)) * 2, int(rng.integers(40, 190)) * 2
    lens = S.gopro_style_lens(w, h)
    lens["f"] = (float(rng.uniform(0.3, 1.2)) * w,) * 2
    lens["c"] = (w / 2.0 + float(rng.uniform(-0.05, 0.05)) * w, h / 2.0 + float(rng.uniform(-0.05, 0.05)) * h)
    lens["k"] = [float(rng.uniform(-0.08, 0.12)), float(rng.uniform(-0.05, 0.05)), float(rng.uniform(-0.03, 0.03)), float(rng.uniform(-0.01, 0.01))] + [0.0] * 8
    fmt = ["NV12", "YUV422P16LE", "YUV420P", "P010LE"][seed % 4]
    return S.SyntheticFrame(fmt, w, h, seed=int(rng.integers(1, 1 << 20)), lens=lens, fov=float(rng.uniform(0.5, 3.0)), readout_ms=float(rng.uniform(-30.0, 30.0)),
                            horizontal_rs=bool(rng.random() < 0.3))


@pytest.mark.parametrize("seed", range(32))
def test_every_certificate_of_a_random_clip(seed):
    fr = random_clip(seed)
    p0 = fr.planes[0]["params"]
    if _emu.p1_table(p0, fr.matrices, p0.matrix_count) is None:
        pytest.skip("no certified first pass for this clip (E >= 0.2 px or a single matrix)")
    outs, a = _emu.run_frames([fr], audit=True)
    assert a["wrong"] == 0 and a["queue_overflow"] == 0 and a["out_of_range"] == 0, a
    # (the bound itself: the lattice form of round 5 spends part of E on the interpolation's curvature, so the measured gap is no longer a small fraction of it)
    assert a["certified"] > 0 and a["gap_px"] < a["eps_px"], a
    assert all(np.array_equal(x, y) for x, y in zip(O.run_frame(fr), outs[0]))


def shifted_frame(shift, fmt="YUV422P16LE", w=640, h=360, seed=3):
    """The same geometry through coordinates far from the origin: translation2d = (shift, shift) and the matrices' constant terms moved the other
    way, so X = ox m0 + oy m1 + m2 is a small difference of large terms — rounding errors of size u * shift * |m0| that a certificate derived
    from the lens alone (rounds 2-3: 1.8e-6 * (|f| rmax smax + |c|) + ...) does not know about."""
    fr = S.SyntheticFrame(fmt, w, h, seed=seed, base_overrides={"translation2d": (shift, shift)})
    m, t = fr.matrices, np.float32(shift)
    for col in (0, 3, 6):
        m[:, col + 2] -= t * m[:, col] + t * m[:, col + 1]
    return fr


def test_cancellation_in_the_matrix_widens_the_certificate():
    fr = shifted_frame(3e4)
    p0 = fr.planes[0]["params"]
    assert _emu.p1_table(p0, fr.matrices, p0.matrix_count) is not None
    outs, a = _emu.run_frames([fr], audit=True)
    assert a["wrong"] == 0 and a["certified"] > 0 and a["gap_px"] < a["eps_px"], a
    # the measured gap is beyond what the lens-only bound of rounds 2-3 allowed: that certificate was unsound for such matrices
    lens_only = 1.5 * 1.2e-6 * (abs(p0.f[1]) * 1.6 + abs(p0.c[1])) + 1.0 / 4096.0
    assert a["gap_px"] > lens_only, (a, lens_only)
    assert all(np.array_equal(x, y) for x, y in zip(O.run_frame(fr), outs[0]))


def test_hopeless_cancellation_certifies_nothing_and_stays_exact():
    fr = shifted_frame(3e5)
    outs, a = _emu.run_frames([fr], audit=True)
    assert a["wrong"] == 0 and a["certified"] == 0, a             # W's own terms dwarf it: every pixel is left to the exact projection
    assert all(np.array_equal(x, y) for x, y in zip(O.run_frame(fr), outs[0]))


def limited_shifted_clip(seed):
    """random clip with an r_limit (cpu_undistort.rs:139: the first pass must agree with the exact path on which pixels have a point at all) and, every other seed,
    coordinates moved away from the origin (the matrix-dependent part of the certificate)"""
    rng = np.random.default_rng(70000 + seed)
    w, h = int(rng.integers(60, 260)) * 2, int(rng.integers(40, 150)) * 2
    lens = S.gopro_style_lens(w, h)
    lens["f"] = (float(rng.uniform(0.35, 1.0)) * w,) * 2
    lens["k"] = [float(rng.uniform(-0.05, 0.1)), float(rng.uniform(-0.04, 0.04)), float(rng.uniform(-0.02, 0.02)), float(rng.uniform(-0.01, 0.01))] + [0.0] * 8
    lens["r_limit"] = float(rng.uniform(0.4, 2.5))
    shift = float(rng.choice([0.0, 300.0, 3000.0, 20000.0])) if seed % 2 else 0.0
    fr = S.SyntheticFrame(["YUV422P16LE", "NV12"][seed % 2], w, h, seed=int(rng.integers(1, 1 << 20)), lens=lens, fov=float(rng.uniform(0.8, 2.5)),
                          readout_ms=float(rng.uniform(-25.0, 25.0)), base_overrides={"translation2d": (shift, -shift)})
    m, t = fr.matrices, np.float32(shift)
    for col in (0, 3, 6):
        m[:, col + 2] -= t * m[:, col] - t * m[:, col + 1]
    return fr


@pytest.mark.parametrize("seed", range(12))
def test_every_certificate_with_an_r_limit_and_shifted_coordinates(seed):
    fr = limited_shifted_clip(seed)
    p0 = fr.planes[0]["params"]
    assert p0.r_limit > 0.0
    if _emu.p1_table(p0, fr.matrices, p0.matrix_count) is None:
        pytest.skip("no certified first pass for this clip")
    outs, a = _emu.run_frames([fr], audit=True)
    assert a["wrong"] == 0 and a["queue_overflow"] == 0 and a["out_of_range"] == 0, a
    if a["certified"]:
        assert a["gap_px"] < a["eps_px"], a
    assert all(np.array_equal(x, y) for x, y in zip(O.run_frame(fr), outs[0]))


# ---- adversarial clips for the derived certificate (round 5) ---------------------------------------------------------------------------------------------------
# Each case aims at one premise of DESIGN.md section 2c: theta_d's polynomial P nearly vanishing inside the table's range (the exact path's relative error explodes: no
# certificate may be issued), a very long and a very short focal length (rho tiny / the table at its 64 cap), a strong roll with a long readout (the lattice's curvature term
# beyond its limit: the frame must fall back to the per-pixel form, not certify wrongly), large coefficients of either sign, a principal point far off centre.  Frames are
# large enough (1280 x 720) that the lattice form is what an ordinary lens gets there, so the adversarial ones are compared with it on its own ground.
def adversarial_clip(case):
    w, h = 1280, 720
    lens = S.gopro_style_lens(w, h)
    kw = dict(seed=900 + case, fov=1.0, readout_ms=16.0)
    if case == 0:      # P(theta) = 1 + k0 t^2 crosses zero near t = 0.95: inside the frame's range
        lens["k"] = [-1.1, 0.0, 0.0, 0.0] + [0.0] * 8
    elif case == 1:    # P with a deep minimum (0.11) inside the range: the exact path's error amplification kappa grows ninefold, E with it; whatever is certified must hold
        lens["k"] = [-1.6, 0.72, 0.0, 0.0] + [0.0] * 8
    elif case == 2:    # very long lens: every ray within a few degrees, rho ~ 1e-3
        lens["f"] = (20.0 * w,) * 2
    elif case == 3:    # very short lens: the corner ray beyond 80 degrees, the table at its cap
        lens["f"] = (0.12 * w,) * 2
        kw["fov"] = 0.5
    elif case == 4:    # strong roll over a long readout: rows of a wave's span far apart, large curvature of v
        kw.update(readout_ms=60.0, constant_quat=None, timestamp_ms=3210.0, fov=2.0)
    elif case == 5:    # large coefficients, alternating signs
        lens["k"] = [0.6, -0.9, 0.7, -0.25] + [0.0] * 8
    elif case == 6:    # principal point far off centre: rho's range lopsided, linear forms with large constant terms
        lens["c"] = (0.1 * w, 0.92 * h)
    elif case == 7:    # horizontal shutter with a zoomed-out frame
        kw.update(horizontal_rs=True, fov=2.5)
    return S.SyntheticFrame("YUV422P16LE", w, h, lens=lens, **kw)


@pytest.mark.parametrize("case", range(8))
def test_adversarial_clips_never_get_a_wrong_certificate(case):
    fr = adversarial_clip(case)
    p0 = fr.planes[0]["params"]
    served = _emu.p1_table(p0, fr.matrices, p0.matrix_count)
    if case == 0:
        assert served is None, "a lens whose theta_d polynomial vanishes inside the range must not be certified"
    if served is None:
        outs = _emu.run_frames([fr])                      # the exact first pass: still the oracle's frame
        assert all(np.array_equal(x, y) for x, y in zip(O.run_frame(fr), outs[0]))
        return
    outs, a = _emu.run_frames([fr], audit=True)
    assert a["wrong"] == 0 and a["queue_overflow"] == 0 and a["out_of_range"] == 0, a
    if a["certified"]:
        assert a["gap_px"] < a["eps_px"], a
    assert all(np.array_equal(x, y) for x, y in zip(O.run_frame(fr), outs[0]))


# ---- round 6: the certified first pass of a radial model other than the fisheye (GoPro's inverted polynomial, gopro.rs:25-72) ------------------------------------
def random_gopro_clip(seed):
    rng = np.random.default_rng(90000 + seed)
    w, h = int(rng.integers(60, 200)) * 2, int(rng.integers(40, 120)) * 2
    lens = S.gopro_style_lens(w, h)
    lens["model"] = "gopro"
    lens["f"] = (float(rng.uniform(0.4, 0.9)) * w,) * 2
    lens["k"] = [0.0, float(rng.uniform(0.9, 1.1)), float(rng.uniform(-0.03, 0.03)), float(rng.uniform(-0.2, 0.02)), float(rng.uniform(-0.04, 0.04)),
                 float(rng.uniform(-0.02, 0.02)), float(rng.uniform(-0.008, 0.008))] + [0.0] * 5
    if seed % 3:
        lens["r_limit"] = float(rng.uniform(1.0, 3.0))
    return S.SyntheticFrame(["YUV422P16LE", "NV12", "YUV420P"][seed % 3], w, h, seed=int(rng.integers(1, 1 << 20)), lens=lens, fov=float(rng.uniform(0.6, 1.8)),
                            readout_ms=float(rng.uniform(-25.0, 25.0)), horizontal_rs=bool(rng.random() < 0.25))


@pytest.mark.parametrize("seed", range(12))
def test_every_certificate_of_a_random_gopro_clip(seed):
    """The table over r and the bounds behind E come from the library's own host code (gfw_debug_p1_radial: the derivation has one statement); what this adds is the
    audit — the interpreted kernel re-derives the exact row of every pixel it certified, the measured |table - exact| must stay inside E — and the oracle."""
    fr = random_gopro_clip(seed)
    if _emu.p1_table_radial(fr) is None:
        pytest.skip("the host declines a certificate for this lens / range (Newton not provably contracting, POLY' too small, E too wide)")
    outs, a = _emu.run_frames([fr], audit=True)
    assert a["wrong"] == 0 and a["queue_overflow"] == 0 and a["out_of_range"] == 0, a
    assert a["certified"] > 0 and a["gap_px"] < a["eps_px"], a
    assert all(np.array_equal(x, y) for x, y in zip(O.run_frame(fr), outs[0]))


CLOSED_FORM_K = {          # coefficient ranges around what the lens databases hold for these models (lensfun's poly3 / poly5 / ptlens; Sony's and the generic theta polynomial)
    "sony": lambda rng: [rng.uniform(0.9, 1.1), rng.uniform(-0.05, 0.05), rng.uniform(-0.15, 0.05), rng.uniform(-0.04, 0.04), rng.uniform(-0.01, 0.01), rng.uniform(-0.004, 0.004)],
    "generic_polynomial": lambda rng: [rng.uniform(0.9, 1.1), rng.uniform(-0.05, 0.05), rng.uniform(-0.15, 0.05), rng.uniform(-0.04, 0.04), rng.uniform(-0.01, 0.01), rng.uniform(-0.004, 0.004)]
                                       + [rng.uniform(-0.001, 0.001) for _ in range(6)],
    "poly3": lambda rng: [rng.uniform(-0.12, 0.12)],
    "poly5": lambda rng: [rng.uniform(-0.12, 0.12), rng.uniform(-0.04, 0.04)],
    "ptlens": lambda rng: [rng.uniform(-0.03, 0.03), rng.uniform(-0.06, 0.06), rng.uniform(-0.04, 0.04)],
}


def random_closed_form_clip(model, seed):
    rng = np.random.default_rng(91000 + 100 * sorted(CLOSED_FORM_K).index(model) + seed)
    w, h = int(rng.integers(60, 200)) * 2, int(rng.integers(40, 120)) * 2
    lens = S.gopro_style_lens(w, h)
    lens["model"] = model
    lens["f"] = (float(rng.uniform(0.4, 0.9)) * w,) * 2
    k = [float(v) for v in CLOSED_FORM_K[model](rng)]
    lens["k"] = k + [0.0] * (12 - len(k))
    if seed % 3:
        lens["r_limit"] = float(rng.uniform(1.0, 3.0))
    return S.SyntheticFrame(["YUV422P16LE", "NV12", "YUV420P"][seed % 3], w, h, seed=int(rng.integers(1, 1 << 20)), lens=lens, fov=float(rng.uniform(0.6, 1.8)),
                            readout_ms=float(rng.uniform(-25.0, 25.0)), horizontal_rs=bool(rng.random() < 0.25))


@pytest.mark.parametrize("seed", range(5))
@pytest.mark.parametrize("model", sorted(CLOSED_FORM_K))
def test_every_certificate_of_a_random_closed_form_radial_clip(model, seed):
    """Sony / generic polynomial (a polynomial in theta = atan r, sony.rs:69-88, generic_polynomial.rs) and lensfun's poly3 / poly5 / ptlens (polynomials in r): no
    iteration in the exact path, so the certificate is the table's error plus the formula's roundings (gfw_api_certificate.inc: p1_prepare_radial_thetapoly / _rpoly).
    The interpreted kernel re-derives the exact row of every pixel it certified; the frame equals the oracle."""
    fr = random_closed_form_clip(model, seed)
    assert _emu.p1_table_radial(fr) is not None, "a closed-form model over a sane range must be certifiable"
    outs, a = _emu.run_frames([fr], audit=True)
    assert a["wrong"] == 0 and a["queue_overflow"] == 0 and a["out_of_range"] == 0, a
    assert a["certified"] > 0 and a["gap_px"] < a["eps_px"], a
    assert all(np.array_equal(x, y) for x, y in zip(O.run_frame(fr), outs[0]))


@pytest.mark.parametrize("model", sorted(CLOSED_FORM_K))
def test_closed_form_tables_against_an_independent_statement(model):
    """The table's entries against the model's formula written here in f64, max |T| and max |T'| against a dense sampling of it: the bounds hold and are not useless;
    the all-zero Sony / generic lens (the reference passes the point through: sony.rs:71) has no table."""
    import ctypes as C
    from gyroflow_amd import abi
    lib = abi.load_library()
    lib.gfw_debug_p1_radial.argtypes = [C.c_void_p, C.c_int, C.c_double, C.c_void_p, C.c_void_p]
    rng = np.random.default_rng(0xC10 + len(model))
    p = abi.KernelParams()
    for i, v in enumerate(CLOSED_FORM_K[model](rng)):
        p.k[i] = v
    k = [float(p.k[i]) for i in range(12)]

    def T(r):
        if model in ("sony", "generic_polynomial"):
            n = 6 if model == "sony" else 12
            t = math.atan(r)
            return k[0] if r == 0.0 else sum(k[i] * t ** (i + 1) for i in range(n)) / r
        if model == "poly3":
            return 1.0 + k[0] * r * r
        if model == "poly5":
            return 1.0 + k[0] * r ** 2 + k[1] * r ** 4
        return 1.0 + k[2] * r + k[1] * r ** 2 + k[0] * r ** 3
    r_max = 2.1
    tab, out = np.zeros((8193, 2), np.float32), np.zeros(7)
    assert lib.gfw_debug_p1_radial(C.byref(p), abi.MODELS[model], r_max, tab.ctypes.data, out.ctypes.data) == 1
    for i in (0, 1, 17, 4096, 8000, 8192):
        assert abs(float(tab[i, 0]) - T(r_max * i / 8192.0)) < 2e-7 * max(1.0, abs(T(r_max * i / 8192.0))), i
        if i < 8192:
            assert abs(float(tab[i, 1]) - (T(r_max * (i + 1) / 8192.0) - T(r_max * i / 8192.0))) < 1e-9, i
    rs = np.linspace(0.0, r_max, 20001)
    ts = np.array([T(r) for r in rs])
    slope = np.abs(np.diff(ts) / np.diff(rs)).max()
    curv = np.abs(np.diff(ts, 2)).max() / (rs[1] - rs[0]) ** 2
    assert np.abs(ts).max() <= out[1] and slope <= out[2] <= 3.0 * slope + 0.05 and curv <= out[3] * (1 + 1e-3) + 1e-3, (np.abs(ts).max(), out[1], slope, out[2], curv, out[3])
    assert out[4] < 1e-6 and out[5] < 1e-5, out                                              # table error and float noise: far below a 32nd of a pixel at any focal length
    if model in ("sony", "generic_polynomial"):
        z = abi.KernelParams()
        z.k[5] = 0.25                                                                       # k0..k3 zero (sony) — the generic model tests all twelve
        assert lib.gfw_debug_p1_radial(C.byref(z), abi.MODELS["sony"], r_max, None, out.ctypes.data) == 0
        assert lib.gfw_debug_p1_radial(C.byref(abi.KernelParams()), abi.MODELS[model], r_max, None, out.ctypes.data) == 0


def test_the_radial_certificate_declines_what_it_cannot_prove():
    import ctypes as C
    from gyroflow_amd import abi
    lib = abi.load_library()
    lib.gfw_debug_p1_radial.argtypes = [C.c_void_p, C.c_int, C.c_double, C.c_void_p, C.c_void_p]
    out = np.zeros(7)

    def ask(k, r_max, model=abi.MODELS["gopro"]):
        p = abi.KernelParams()
        for i, v in enumerate(k):
            p.k[i] = v
        return lib.gfw_debug_p1_radial(C.byref(p), model, r_max, None, out.ctypes.data)
    good = [0.0, 1.0, 0.01, -0.12, 0.02, 0.01, -0.004]
    assert ask(good, 1.5) == 1 and out[6] > 0.5 and 0.9 < out[1] < 1.1                       # d_min, Tmax of a real lens
    assert ask([0.01] + good[1:], 1.5) == 0                                                 # k0 != 0: the scale has a pole at the optical centre
    assert ask([0.0, 1.0, 0.0, -0.5, 0.0, 0.0, 0.0], 1.5) == 0                               # POLY' = 1 - 1.5 p^2 vanishes inside the range: the map folds
    assert ask(good, 9.0) == 0                                                              # beyond the range the certificate is written for
    assert ask(good, 1.5, model=abi.MODELS["opencv_fisheye"]) < 0                           # not a radial-table model


def test_the_radial_table_and_its_bounds_against_an_independent_statement():
    """T(r) = k1 q(atan r) / r with q = POLY^-1: the table's entries against a bisection in f64 written here, and the bounds the certificate uses (max T, max |T'|)
    against a dense sampling of the same statement — the bounds must hold, and not be useless."""
    import ctypes as C
    from gyroflow_amd import abi
    lib = abi.load_library()
    lib.gfw_debug_p1_radial.argtypes = [C.c_void_p, C.c_int, C.c_double, C.c_void_p, C.c_void_p]
    k = [0.0, 1.03, -0.02, -0.15, 0.03, 0.012, -0.005]
    p = abi.KernelParams()
    for i, v in enumerate(k):
        p.k[i] = v
    kf = [float(p.k[i]) for i in range(7)]                                                  # the coefficients as the kernel sees them (f32)
    tab, out = np.zeros((8193, 2), np.float32), np.zeros(7)
    r_max = 1.7
    assert lib.gfw_debug_p1_radial(C.byref(p), abi.MODELS["gopro"], r_max, tab.ctypes.data, out.ctypes.data) == 1

    def q(theta):
        lo, hi = 0.0, 4.0
        for _ in range(200):
            mid = 0.5 * (lo + hi)
            if sum(c * mid ** i for i, c in enumerate(kf)) < theta:
                lo = mid
            else:
                hi = mid
        return 0.5 * (lo + hi)

    def T(r):
        return 1.0 if r == 0.0 else kf[1] * q(np.arctan(r)) / r
    for i in (0, 1, 17, 4096, 8000, 8192):
        assert abs(float(tab[i, 0]) - T(r_max * i / 8192.0)) < 2e-7, i
    rs = np.linspace(0.0, r_max, 4001)
    ts = np.array([T(r) for r in rs])
    slope = np.abs(np.diff(ts) / np.diff(rs)).max()
    assert ts.max() <= out[1] and slope <= out[2] <= 3.0 * slope + 0.05, (ts.max(), out[1], slope, out[2])
