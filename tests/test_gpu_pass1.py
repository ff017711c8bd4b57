"""Audit of the fused kernel's certified first pass (gfw_frame.hip): in audit mode the kernel recomputes the exact
rolling-shutter row for EVERY pixel whose approximate row was accepted and counts disagreements — there must be none —
and reports how many pixels were queued to the exact path."""
import numpy as np
import pytest

from gyroflow_amd import abi, synthetic as S, warp
import _oracle as O

pytestmark = pytest.mark.gpu


def eps_of(fr):
    """A floor of the certificate half-width (gfw_api.hip: p1_setup; gfw_frame.hip: p1_bound): never below 2^-14 px + 2 u (|c| + |f|) * 10."""
    p = fr.planes[0]["params"]
    return 1.0 / 16384.0 + 1.2e-6 * (abs(p.c[1]) + abs(p.f[1]))


LAST_FULL = {}


def audit(fr, variant=3):
    """variant 3: the fused kernel's audit instantiation (first pass)."""
    outs = [pl["dst"].copy() for pl in fr.planes]
    bufs = [warp.host_buffers(pl["src"], pl["size"], o, pl["out_size"]) for pl, o in zip(fr.planes, outs)]
    params = [pl["params"] for pl in fr.planes]
    types = [pl["pixel_type"] for pl in fr.planes]
    be = warp.Backend(params[0], types[0], fr.model, fr.digital, bufs[0])
    try:
        be.set_option(abi.OPT_KERNEL_VARIANT, variant)
        be.get_audit(reset=True)
        be.undistort_frame(bufs, params, types, fr.matrices)
        assert warp.last_backend().startswith("yuv_fused_p1")
        full = be.get_audit_full()
        assert full["out_of_range"] == 0, full
        LAST_FULL.clear(); LAST_FULL.update(full)
        return be.get_audit(), outs
    finally:
        be.close()


@pytest.mark.parametrize("seed", [1, 2, 3, 4, 5, 6])
@pytest.mark.parametrize("size", [(640, 360), (1920, 1080)])
def test_certificates_never_disagree_with_the_exact_row(seed, size):
    fr = S.SyntheticFrame("YUV422P16LE", size[0], size[1], seed=seed, timestamp_ms=500.0 + 77.7 * seed, readout_ms=8.0 + 4.0 * seed)
    (certified, wrong, queued, overflow, gap), outs = audit(fr)
    total = size[0] * size[1]
    assert certified + queued + overflow == total
    assert wrong == 0
    assert gap < LAST_FULL["pass1_eps_px"], "approximation gap %g px exceeds the certificate half-width %g" % (gap, LAST_FULL["pass1_eps_px"])
    # the per-pixel form (rounds 2-4) keeps its margin: the gap is rounding noise, far inside even the floor of E
    (certified, wrong, queued, overflow, gap), outs2 = audit(fr, variant=4)
    assert certified + queued + overflow == total and wrong == 0
    assert gap < 0.5 * eps_of(fr), "approximation gap %g px is not well inside the certificate half-width" % gap
    for a, b in zip(outs, outs2):
        assert np.array_equal(a, b)
    assert queued + overflow < 0.15 * total, "certificate rejects too many pixels: %d of %d" % (queued + overflow, total)
    ref = O.run_frame(fr)
    for a, b in zip(ref, outs):
        assert np.array_equal(a, b)


def test_audit_4k_c2_and_wide_lens():
    fr = S.SyntheticFrame("YUV422P16LE", 3840, 2160, seed=0x9F10)
    (certified, wrong, queued, overflow, gap), _ = audit(fr)
    assert wrong == 0 and certified > 0.85 * 3840 * 2160
    lens = S.gopro_style_lens(1280, 720)
    lens["f"] = (0.33 * 1280, 0.33 * 1280)                      # much wider field of view: rho up to ~3.5
    lens["k"] = [0.12, -0.04, 0.01, -0.002] + [0.0] * 8
    fr = S.SyntheticFrame("NV12", 1280, 720, seed=8, lens=lens, fov=1.3)
    (certified, wrong, queued, overflow, gap), outs = audit(fr)
    assert wrong == 0
    ref = O.run_frame(fr)
    for a, b in zip(ref, outs):
        assert np.array_equal(a, b)


def test_horizontal_rs_and_zoomed_out_audit():
    fr = S.SyntheticFrame("YUV422P16LE", 640, 360, seed=4, horizontal_rs=True)
    (certified, wrong, queued, overflow, gap), _ = audit(fr)
    assert wrong == 0 and certified > 0
    fr = S.SyntheticFrame("YUV422P16LE", 640, 360, seed=17, fov=3.0)
    (certified, wrong, queued, overflow, gap), outs = audit(fr)
    assert wrong == 0
    ref = O.run_frame(fr)
    for a, b in zip(ref, outs):
        assert np.array_equal(a, b)


def test_all_zero_k_with_rolling_shutter_keeps_the_certified_pass_exact():
    # opencv_fisheye with k[0..3] == 0 (no lens profile / rectilinear): the reference returns (x/z, y/z) without the atan
    # scaling (opencv_fisheye.rs:75); the first-pass table must describe that very map or rows are picked tens of pixels off
    lens = S.gopro_style_lens(960, 540)
    lens["k"] = [0.0] * 12
    for fmt, seed in (("YUV422P16LE", 31), ("NV12", 32)):
        fr = S.SyntheticFrame(fmt, 960, 540, seed=seed, lens=dict(lens), readout_ms=20.0)
        assert fr.matrices.shape[0] == 540
        (certified, wrong, queued, overflow, gap), outs = audit(fr)
        assert wrong == 0 and certified > 0.8 * 960 * 540
        ref = O.run_frame(fr)
        for a, b in zip(ref, outs):
            assert np.array_equal(a, b)
        got = warp.run_frame(fr)
        assert warp.last_backend() == "yuv_fused_p1"
        for a, b in zip(ref, got):
            assert np.array_equal(a, b)


def test_the_kernel_widens_the_certificate_for_a_matrix_with_cancellation():
    """translation2d = (3e4, 3e4) with the matrices' constant terms moved the other way: the linear forms are small differences of large terms.  The kernel
    derives E from the matrix it uses (device-resident tables included), so the reported E grows and every certificate still holds; the measured gap is
    beyond the lens-only bound of rounds 2-3."""
    from test_emu_pass1_audit import shifted_frame
    fr = shifted_frame(3e4, w=1280, h=720)
    be_audit, outs = audit(fr, variant=4)          # (the per-pixel form: its gap is rounding noise alone; the lattice form's includes the interpolation's)
    certified, wrong, queued, overflow, gap = be_audit
    assert wrong == 0 and certified > 0
    eps_shifted = LAST_FULL["pass1_eps_px"]
    assert gap < eps_shifted
    plain = S.SyntheticFrame("YUV422P16LE", 1280, 720, seed=3)
    (_, _, _, _, gap_plain), _ = audit(plain, variant=4)
    assert gap > 5.0 * gap_plain and eps_shifted > 3.0 * LAST_FULL["pass1_eps_px"]
    ref = O.run_frame(fr)
    for a, b in zip(ref, outs):
        assert np.array_equal(a, b)
    # device-resident matrices: the host has no view of them, the kernel's own E decides
    got = warp.run_frame(fr)
    for a, b in zip(ref, got):
        assert np.array_equal(a, b)
    # the lattice form on the same frame: a wider E, every certificate still right
    (certified, wrong, queued, overflow, gap), outs3 = audit(fr)
    assert wrong == 0 and certified > 0 and gap < LAST_FULL["pass1_eps_px"]
    for a, b in zip(ref, outs3):
        assert np.array_equal(a, b)
    fr = shifted_frame(3e5, w=1280, h=720)
    (certified, wrong, queued, overflow, gap), outs = audit(fr)
    assert certified == 0 and wrong == 0
    for a, b in zip(O.run_frame(fr), outs):
        assert np.array_equal(a, b)
