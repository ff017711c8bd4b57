"""Inverse point map on the device (gfw_undistort_points) vs the oracle restatement of `undistort_points`
(cpu_undistort.rs:652-858) — bit-exact, every lens model, digital lenses, refraction, stretch, IBIS/OIS shifts, Sony mesh,
per-point / per-row / per-column rotation rows, and the STMap "dist" grid walk (stmap.rs:123-127)."""
import numpy as np
import pytest

from gyroflow_amd import abi, synthetic as S, warp
import _oracle as O
from test_gpu_lens_models import PHYSICAL, DIGITAL, synthetic_mesh
from test_oracle_points import points_params

pytestmark = pytest.mark.gpu


def backend_for(fr):
    pl = fr.planes[0]
    b = warp.host_buffers(pl["src"], pl["size"], pl["dst"].copy(), pl["out_size"])
    return warp.Backend(pl["params"], pl["pixel_type"], fr.model, fr.digital, b)


def wild_points(w, h, n, seed):
    rng = np.random.default_rng(seed)
    pts = np.stack([rng.uniform(-0.5 * w, 1.5 * w, n), rng.uniform(-0.5 * h, 1.5 * h, n)], axis=1).astype(np.float32)
    pts[:8] = [[0, 0], [w, h], [w / 2, h / 2], [1e7, 1e7], [-1e6, 3.0], [np.nan, 1.0], [np.inf, 0.0], [w / 2 + 1e-3, h / 2]]
    return pts


def same_bits(a, b):
    """Bit-identical, except that NaN compares equal to NaN (sign/payload of an invalid-operation NaN is an ISA detail:
    x86 produces 0xFFC00000, gfx950 0x7FC00000; Rust does not specify it either)."""
    a, b = np.asarray(a), np.asarray(b)
    eq = (a.view(np.uint32) == b.view(np.uint32)) | (np.isnan(a) & np.isnan(b))
    if not eq.all():
        bad = np.argwhere(~eq)
        print("first mismatches:", [(tuple(i), a[tuple(i)], b[tuple(i)]) for i in bad[:5]], "of", len(bad))
    return bool(eq.all())


@pytest.mark.parametrize("model", sorted(PHYSICAL))
def test_points_every_lens_model_bit_exact(model):
    w, h = 320, 200
    lens = S.gopro_style_lens(w, h)
    lens["model"] = model
    lens["k"] = PHYSICAL[model] + [0.0] * (12 - len(PHYSICAL[model]))
    fr = S.SyntheticFrame("YUV422P16LE", w, h, seed=61, lens=lens, fov=1.3)
    kp = points_params(fr)
    pts = wild_points(w, h, 20000, 7)
    rows = np.clip(np.round(pts[:, 1]), 0, h - 1)
    rows = np.nan_to_num(rows, nan=0.0, posinf=h - 1, neginf=0).astype(np.int64)
    rot = fr.rotations[rows]
    ref = O.undistort_points(kp, fr.model, 0, rot, points=pts, index_mode=abi.POINT_INDEX_PER_POINT)
    be = backend_for(fr)
    try:
        got = be.undistort_points(kp, rot, points=pts, index_mode=abi.POINT_INDEX_PER_POINT)
        assert warp.last_backend() == "points"
    finally:
        be.close()
    assert same_bits(ref, got)
    assert np.isfinite(ref).mean() > 0.9


@pytest.mark.parametrize("digital", sorted(DIGITAL))
def test_points_digital_lens_refraction_stretch(digital):
    w, h = 256, 144
    lens = S.gopro_style_lens(w, h)
    lens["digital"] = digital
    fr = S.SyntheticFrame("NV12", w, h, seed=67, lens=lens, fov=1.1, base_overrides={"digital_lens_params": DIGITAL[digital]})
    kp = points_params(fr)
    kp.light_refraction_coefficient = 1.33
    kp.input_horizontal_stretch, kp.input_vertical_stretch = 1.2, 0.9
    pts = wild_points(w, h, 8000, 11)
    ref = O.undistort_points(kp, fr.model, fr.digital, fr.rotations, points=pts, index_mode=abi.POINT_INDEX_SINGLE)
    be = backend_for(fr)
    try:
        got = be.undistort_points(kp, fr.rotations, points=pts, index_mode=abi.POINT_INDEX_SINGLE)
    finally:
        be.close()
    assert same_bits(ref, got)


@pytest.mark.parametrize("hrs", [False, True])
def test_stmap_dist_grid_with_shifts_bit_exact(hrs):
    """The STMap 'dist' pass: every pixel of the source grid, rotation row picked by the pixel's own row / column."""
    w, h = 384, 216
    fr = S.SyntheticFrame("YUV422P16LE", w, h, seed=71, fov=1.2, horizontal_rs=hrs)
    kp = points_params(fr)
    n = fr.rotations.shape[0]
    t = np.arange(n, dtype=np.float64)
    shifts = np.stack([1.5 * np.sin(t / 37.0), -0.8 * np.cos(t / 23.0), 0.004 * np.sin(t / 51.0), 0.3 * np.cos(t / 19.0), 0.2 * np.sin(t / 29.0)], axis=1).astype(np.float32)
    mode = abi.POINT_INDEX_PER_COLUMN if hrs else abi.POINT_INDEX_PER_ROW
    be = backend_for(fr)
    try:
        for sh in (None, shifts):
            ref = O.undistort_points(kp, fr.model, 0, fr.rotations, grid=(w, h), shifts=sh, index_mode=mode)
            got = be.undistort_points(kp, fr.rotations, grid=(w, h), shifts=sh, index_mode=mode)
            assert same_bits(ref, got)
            # the centre of the source image lands near the centre of the output
            assert abs(got[h // 2, w // 2, 0] - w / 2) < 0.2 * w and abs(got[h // 2, w // 2, 1] - h / 2) < 0.2 * h
    finally:
        be.close()


@pytest.mark.parametrize("with_mesh,with_fpd", [(True, False), (True, True), (False, True)])
def test_points_sony_mesh_f64(with_mesh, with_fpd):
    w, h = 256, 160
    lens = S.gopro_style_lens(w, h)
    lens["model"] = "sony"
    lens["k"] = PHYSICAL["sony"] + [0.0] * 6
    fr = S.SyntheticFrame("YUV422P16LE", w, h, seed=73, lens=lens, fov=1.2)
    kp = points_params(fr)
    mesh = synthetic_mesh(w, h, with_fpd, with_mesh).astype(np.float64)
    mesh[20:40] += 1e-11                                      # values that are not f32-representable: the inverse takes the f64 mesh as is
    ref = O.undistort_points(kp, fr.model, 0, fr.rotations, grid=(w, h), index_mode=abi.POINT_INDEX_PER_ROW, mesh=mesh)
    be = backend_for(fr)
    try:
        got = be.undistort_points(kp, fr.rotations, grid=(w, h), index_mode=abi.POINT_INDEX_PER_ROW, mesh=mesh)
    finally:
        be.close()
    assert same_bits(ref, got)


@pytest.mark.parametrize("model", ["opencv_fisheye", "opencv_standard", "poly5", "sony", "gopro", "insta360"])
@pytest.mark.parametrize("lca,digital", [(0.35, None), (0.8, "gopro_superview"), (0.0, "digital_stretch")])
def test_points_lens_correction_branch_bit_exact(model, lca, digital):
    """lens_correction_amount < 1: the Newton inverse of the render blend (cpu_undistort.rs:785-851) on the device."""
    w, h = 256, 144
    lens = S.gopro_style_lens(w, h)
    lens["model"] = model
    lens["k"] = PHYSICAL[model] + [0.0] * (12 - len(PHYSICAL[model]))
    ov = {"lens_correction_amount": lca}
    if digital:
        lens["digital"] = digital
        ov["digital_lens_params"] = DIGITAL[digital]
    fr = S.SyntheticFrame("YUV422P16LE", w, h, seed=83, lens=lens, fov=1.25, base_overrides=ov)
    kp = points_params(fr)
    kp.lens_correction_amount = lca
    kp.fov = fr.planes[0]["params"].fov
    kp.light_refraction_coefficient = 1.0 if model != "sony" else 1.33
    pts = wild_points(w, h, 6000, 13)
    rows = np.nan_to_num(np.clip(np.round(pts[:, 1]), 0, h - 1), nan=0.0, posinf=h - 1, neginf=0).astype(np.int64)
    rot = fr.rotations[rows]
    ref = O.undistort_points(kp, fr.model, fr.digital, rot, points=pts, index_mode=abi.POINT_INDEX_PER_POINT)
    be = backend_for(fr)
    try:
        got = be.undistort_points(kp, rot, points=pts, index_mode=abi.POINT_INDEX_PER_POINT)
    finally:
        be.close()
    assert same_bits(ref, got)


def test_points_argument_errors():
    w, h = 64, 48
    fr = S.SyntheticFrame("NV12", w, h, seed=3)
    kp = points_params(fr)
    be = backend_for(fr)
    try:
        assert be.undistort_points(kp, fr.rotations, points=np.zeros((0, 2), np.float32)).shape == (0, 2)
        with pytest.raises(warp.GfwError) as e:
            be.undistort_points(kp, fr.rotations, points=np.ones((4, 2), np.float32), index_mode=9)
        assert e.value.code == abi.ERR_INVALID_ARGUMENT
    finally:
        be.close()
