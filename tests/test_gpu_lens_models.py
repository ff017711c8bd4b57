"""Every physical lens model and digital lens (distortion_models/*.rs) through the generic kernel, both directions:
lens_correction_amount = 1 exercises distort_point, < 1 additionally exercises undistort_point (cpu_undistort.rs:429-460)."""
import numpy as np
import pytest

from gyroflow_amd import abi, synthetic as S, warp
import _oracle as O
from test_gpu_parity import assert_plane_equal

pytestmark = pytest.mark.gpu

PHYSICAL = {
    "opencv_fisheye": [0.045, 0.02, -0.02, 0.006],
    "opencv_standard": [0.12, -0.05, 0.001, 0.002, 0.01, 0.02, -0.01, 0.001, 0.0005, -0.0002, 0.0003, 0.0001],
    "poly3": [0.06],
    "poly5": [0.08, -0.02],
    "ptlens": [0.01, -0.03, 0.02],
    "insta360": [0.05, -0.01, 0.002, 0.001, -0.001, 0.6],
    "sony": [1.0, 0.01, -0.05, 0.02, 0.003, -0.001],
    "generic_polynomial": [1.0, 0.01, -0.05, 0.02, 0.003, -0.001, 0.0005, 0.0, 0.0, 0.0, 0.0, 0.0],
    "gopro": [0.0, 1.0, 0.01, -0.12, 0.02, 0.01, -0.004],
}
DIGITAL = {
    "gopro_superview": [],
    "gopro6_superview": [],
    "gopro_hyperview": [],
    "digital_stretch": [1.1, 0.95],
    "gopro_warp": [1.32, -1.2, 1.6, -0.4, 0.1, 0.0, 0.0, -0.1, 0.95, 0.4, -0.7, -0.35, 1.1, 0.35, 1.3333334],
}


def run(fr):
    ref = O.run_frame(fr)
    got = warp.run_frame(fr)
    for i, (a, b) in enumerate(zip(ref, got)):
        assert_plane_equal(a, b, fr.planes[i]["pixel_type"], "%s plane %d" % (warp.last_backend(), i))


@pytest.mark.parametrize("model", sorted(PHYSICAL))
@pytest.mark.parametrize("lca", [1.0, 0.45])
def test_physical_lens_models(model, lca):
    w, h = 256, 160
    lens = S.gopro_style_lens(w, h)
    lens["model"] = model
    lens["k"] = PHYSICAL[model] + [0.0] * (12 - len(PHYSICAL[model]))
    if model == "gopro":
        lens["r_limit"] = 2.5
    fr = S.SyntheticFrame("YUV422P16LE", w, h, seed=31, lens=lens, fov=1.2, base_overrides={"lens_correction_amount": lca})
    run(fr)
    assert warp.last_backend().startswith("yuv_fused")  # every physical model, with and without the lens-correction blend, runs fused


@pytest.mark.parametrize("digital", sorted(DIGITAL))
@pytest.mark.parametrize("lca", [1.0, 0.6])
def test_digital_lenses(digital, lca):
    w, h = 256, 144
    lens = S.gopro_style_lens(w, h)
    lens["digital"] = digital
    fr = S.SyntheticFrame("NV12", w, h, seed=37, lens=lens, fov=1.1,
                          base_overrides={"lens_correction_amount": lca, "digital_lens_params": DIGITAL[digital]})
    assert fr.planes[0]["params"].flags & abi.FLAG_HAS_DIGITAL_LENS
    run(fr)
    assert warp.last_backend() == "yuv_fused"           # digital lenses and the lens-correction blend run fused (generic-model instantiation)


def test_r_limit_and_stretch_and_input_rotation():
    w, h = 256, 160
    lens = S.gopro_style_lens(w, h)
    lens["r_limit"] = 0.9
    run(S.SyntheticFrame("YUV422P16LE", w, h, seed=41, lens=lens, fov=1.6))
    run(S.SyntheticFrame("YUV422P16LE", w, h, seed=41, fov=1.1, base_overrides={"input_horizontal_stretch": 1.25, "input_vertical_stretch": 0.9}))
    fr = S.SyntheticFrame("NV12", w, h, seed=43, fov=1.3)
    for pl in fr.planes:
        pl["params"].input_rotation = 90.0
    run(fr)
    fr = S.SyntheticFrame("YUV422P16LE", w, h, seed=43, fov=1.3)
    for pl in fr.planes:
        pl["params"].input_rotation = 17.5
    run(fr)


def synthetic_mesh(w, h, with_fpd=True, with_mesh=True):
    """A Sony-style mesh_data block (gyro_source/sony.rs, splines.rs:88-177 layout): header[9], 9x9 raw grid, per-row
    cubic coefficient blocks for x and y, 20 floats of focal-plane-distortion data."""
    n = 9
    m = np.zeros(839, dtype=np.float32)
    o = 9 + n * n * 2 + n * n * 4 * 2
    m[0] = o if with_mesh else 5.0          # > 10 enables the mesh; <= 10 but > 0 leaves only the FPD branch reachable
    if not with_mesh:
        o = 5
        m = np.zeros(839, dtype=np.float32); m[0] = o
    m[1], m[2], m[3], m[4] = n, n, w, h
    m[5], m[6], m[7], m[8] = 0.0, 0.0, w, h
    if with_mesh:
        base = 9 + n * n * 2
        for comp in range(2):
            for j in range(n):
                rb = base + comp * n * n * 4 + j * n * 4
                for i in range(n):
                    if comp == 0:
                        a = i * w / 8.0 + 1.5 * np.sin(0.7 * i + 0.3 * j); b = 1.0 + 0.01 * np.cos(i + j); c = 1e-4 * (i - 4); d = -1e-7 * (j - 3)
                    else:
                        a = j * h / 8.0 + 1.2 * np.cos(0.5 * i - 0.2 * j); b = 0.004 * np.sin(i - j); c = 2e-5 * (j - 4); d = 1e-8 * (i - 2)
                    m[rb + i], m[rb + n + i], m[rb + 2 * n + i], m[rb + 3 * n + i] = a, b, c, d
    if with_fpd:
        m[o] = 1.0
        for idx in range(8):
            m[o + 4 + idx * 2 + 0] = 0.002 * (idx - 3)
            m[o + 4 + idx * 2 + 1] = -0.001 * (idx - 4)
    return m


def run_with_mesh(fr, mesh):
    ref = []
    for pl in fr.planes:
        dst = pl["dst"].copy()
        assert O.undistort_image(pl["src"], pl["size"], dst, pl["out_size"], pl["params"], pl["pixel_type"], fr.model, fr.digital, fr.matrices, mesh=mesh) == 1
        ref.append(dst)
    for pl, r in zip(fr.planes, ref):
        dst = pl["dst"].copy()
        b = warp.host_buffers(pl["src"], pl["size"], dst, pl["out_size"])
        be = warp.Backend(pl["params"], pl["pixel_type"], fr.model, fr.digital, b)
        try:
            be.undistort_image(b, pl["params"], fr.matrices, mesh)
        finally:
            be.close()
        assert_plane_equal(r, dst, pl["pixel_type"], "mesh")


@pytest.mark.parametrize("with_mesh,with_fpd,inverted", [(True, False, False), (True, True, False), (True, True, True), (False, True, False)])
def test_sony_mesh_and_focal_plane_distortion(with_mesh, with_fpd, inverted):
    w, h = 192, 128
    fr = S.SyntheticFrame("NV12", w, h, seed=47, fov=1.1, flags=abi.FLAG_FRAMEBUFFER_INVERTED if inverted else 0)
    run_with_mesh(fr, synthetic_mesh(w, h, with_fpd, with_mesh))


def test_ibis_ois_terms_in_matrices():
    w, h = 256, 160
    fr = S.SyntheticFrame("YUV422P16LE", w, h, seed=53, fov=1.2)
    y = np.arange(fr.matrices.shape[0], dtype=np.float32)
    fr.matrices[:, 9] = 1.5 * np.sin(y * 0.05)          # sx
    fr.matrices[:, 10] = -0.8 * np.cos(y * 0.03)        # sy
    fr.matrices[:, 11] = 0.004 * np.sin(y * 0.02)       # roll angle (radians)
    fr.matrices[:, 12] = 0.6                            # ox
    fr.matrices[:, 13] = -0.4                           # oy
    ref = O.run_frame(fr)
    got = warp.run_frame(fr)
    assert warp.last_backend() == "yuv_fused"           # served by the fused kernel's generic-model instantiation (exact first pass)
    gen = warp.run_frame(fr, fused=False)
    assert warp.last_backend() == "plane_generic"
    for i, (a, b, g) in enumerate(zip(ref, got, gen)):
        assert_plane_equal(a, b, fr.planes[i]["pixel_type"], "ibis plane %d (fused)" % i)
        assert_plane_equal(a, g, fr.planes[i]["pixel_type"], "ibis plane %d (generic)" % i)


@pytest.mark.parametrize("mode", [1, 2])
def test_ibis_terms_with_device_resident_matrices(mode):
    """Raw rows[14] (mode 1: cos/sin of the roll evaluated on the device with the restated libm routines) and packed
    rows[16] (mode 2) resident in HBM, GFW_FLAG_HAS_IBIS_DATA set as get_kernel_flags does (mod.rs:226-251)."""
    import torch
    w, h = 256, 160
    fr = S.SyntheticFrame("YUV422P16LE", w, h, seed=59, fov=1.2, flags=abi.FLAG_HAS_IBIS_DATA)
    y = np.arange(fr.matrices.shape[0], dtype=np.float32)
    fr.matrices[:, 9] = 1.25 * np.sin(y * 0.05)
    fr.matrices[:, 10] = -0.7 * np.cos(y * 0.03)
    fr.matrices[:, 11] = 0.05 * np.sin(y * 0.02) + 0.01     # roll angle (radians), large enough to matter
    fr.matrices[:, 12] = 0.5
    fr.matrices[:, 13] = -0.25
    fr.matrices[::7, 9:14] = 0.0                             # some rows without IBIS data: cos/sin slots must be 1/0 there
    ref = O.run_frame(fr)
    dev = torch.device("cuda", 0)
    host = fr.matrices if mode == 1 else warp.pack_matrices(fr.matrices)
    d_mat = torch.from_numpy(np.ascontiguousarray(host)).to(dev)
    torch.cuda.synchronize(dev)                              # (the table is uploaded on torch's stream, read on the context's)
    for variant in (0, 1):                                   # fused (generic-model instantiation) and per-plane kernels
        outs = [pl["dst"].copy() for pl in fr.planes]
        bufs = [warp.host_buffers(pl["src"], pl["size"], o, pl["out_size"]) for pl, o in zip(fr.planes, outs)]
        params = [pl["params"] for pl in fr.planes]
        types = [pl["pixel_type"] for pl in fr.planes]
        be = warp.Backend(params[0], types[0], fr.model, fr.digital, bufs[0])
        try:
            be.set_option(abi.OPT_MATRICES_ON_DEVICE, mode)
            if variant:
                be.set_option(abi.OPT_KERNEL_VARIANT, variant)
            be.undistort_frame(bufs, params, types, d_mat.data_ptr(), matrix_count=fr.matrices.shape[0])
            assert warp.last_backend() == ("plane_generic" if variant else "yuv_fused")
        finally:
            be.close()
        for i, (a, b) in enumerate(zip(ref, outs)):
            assert_plane_equal(a, b, fr.planes[i]["pixel_type"], "ibis mode %d variant %d plane %d" % (mode, variant, i))


@pytest.mark.parametrize("model", sorted(PHYSICAL))
@pytest.mark.parametrize("fmt,lca", [("YUV422P16LE", 1.0), ("NV12", 1.0), ("YUV420P", 0.45)])
def test_every_frame_of_a_clip_launch_for_every_lens_model(model, fmt, lca):
    """gfw_undistort_clip, four frames in one launch of the clip's specialised build, device-resident tables: every frame lands in its own planes and equals the
    oracle's — for every lens model (round 6: one model's build took a clip launch for a single frame, tests/test_gpu_pass1_radial.py tells the story; the frame-by-
    frame tests above could not see it)."""
    import test_gpu_jit as J
    w, h = 384, 216
    lens = S.gopro_style_lens(w, h)
    lens["model"] = model
    lens["k"] = PHYSICAL[model] + [0.0] * (12 - len(PHYSICAL[model]))
    if model == "gopro":
        lens["r_limit"] = 2.5
    frames = [S.SyntheticFrame(fmt, w, h, seed=0x9F10 + j, timestamp_ms=1000.0 + 33.3 * j, lens=dict(lens), fov=1.2, readout_ms=16.0, pixels=False,
                               base_overrides={"lens_correction_amount": lca}) for j in range(4)]
    backend, status, (ms, launches, covered), outs, srcs = J.device_clip(frames, 2, True)
    assert backend.endswith("_jit") and status[0] == 2 and launches == 1 and covered == 4, (backend, status, launches, covered)
    for j, fr in enumerate(frames):
        for p, (a, b) in enumerate(zip(O.run_frame(J._View(fr, srcs[j])), outs[j])):
            assert_plane_equal(a, b, fr.planes[p]["pixel_type"], "%s clip launch (%s), frame %d plane %d" % (model, backend, j, p))
