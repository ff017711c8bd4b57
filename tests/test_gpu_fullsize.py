"""Full-size BASELINE configurations through the path bench.py uses.

C2 (3840x2160 YUV422P16LE), C3 (7680x4320), C4 (4K RGBAF32 and GBRAPF32LE with the adaptive-zoom crop) — source
frames produced in HBM, HIP_DEVICE buffers, packed per-row matrices resident on the device
(GFW_OPT_MATRICES_ON_DEVICE = 2), asynchronous calls on the caller's (torch) stream — compared plane for plane with the
oracle.  Bar (BASELINE.json north_star): bit-exact for u8/u16, <= 1 ULP for f32.  Reference loop:
cpu_undistort.rs:519-626.
"""
import ctypes as C

import numpy as np
import pytest

from gyroflow_amd import abi, synthetic as S, warp
import _oracle as O
from test_gpu_parity import assert_plane_equal

pytestmark = pytest.mark.gpu


class _View:
    def __init__(self, frame, src):
        self.model, self.digital, self.matrices = frame.model, frame.digital, frame.matrices
        self.planes = []
        for pl, s in zip(frame.planes, src):
            q = dict(pl)
            q["src"] = s
            q["dst"] = np.full(pl["out_size"][2] * pl["out_size"][1], 0x5A, dtype=np.uint8)
            self.planes.append(q)


def run_device_path(frames, stream_kind="torch", variant=0, audit_out=None):
    """Warp every frame the way bench.py does; returns [(frame, [src planes on host], [dst planes on host])]."""
    import torch
    dev = torch.device("cuda", 0)
    lib = abi.load_library()
    assert lib.gfw_set_device(0) == 0
    side = torch.cuda.Stream(dev) if stream_kind == "side" else None
    stream = side or torch.cuda.current_stream(dev)
    d_src = [fr.device_planes(dev) for fr in frames]
    d_dst = [fr.device_outputs(dev) for fr in frames]
    d_mat = [torch.from_numpy(warp.pack_matrices(fr.matrices)).to(dev) for fr in frames]
    torch.cuda.synchronize(dev)
    types = [pl["pixel_type"] for pl in frames[0].planes]
    params = [[pl["params"] for pl in fr.planes] for fr in frames]
    bufs = [[warp.device_buffers(d_src[j][p].data_ptr(), d_src[j][p].numel(), pl["size"], d_dst[j][p].data_ptr(), d_dst[j][p].numel(), pl["out_size"])
             for p, pl in enumerate(fr.planes)] for j, fr in enumerate(frames)]
    be = warp.Backend(params[0][0], types[0], frames[0].model, frames[0].digital, bufs[0][0])
    try:
        be.set_stream(stream.cuda_stream)
        be.set_option(abi.OPT_SYNCHRONOUS, 0)
        be.set_option(abi.OPT_MATRICES_ON_DEVICE, 2)
        if variant:
            be.set_option(abi.OPT_KERNEL_VARIANT, variant)
            be.get_audit(reset=True)
        calls = [warp.FrameCall(be, bufs[j], params[j], types, d_mat[j].data_ptr(), frames[j].matrices.shape[0]) for j in range(len(frames))]
        for rep in range(2):                              # twice: steady state, all frames in flight back to back
            for c in calls:
                c()
        be.synchronize()
        backend = warp.last_backend()
        if audit_out is not None:
            arr = (C.c_ulonglong * 8)()
            assert be.lib.gfw_get_audit(be.ctx, C.byref(arr), 0) == 0
            audit_out.extend(int(v) for v in arr)
    finally:
        be.close()
    torch.cuda.synchronize(dev)
    return backend, [(fr, [t.cpu().numpy() for t in d_src[j]], [t.cpu().numpy() for t in d_dst[j]]) for j, fr in enumerate(frames)]


def check(frames, expect_backend, stream_kind="torch"):
    backend, res = run_device_path(frames, stream_kind)
    assert backend.startswith(expect_backend), backend
    for fr, src, dst in res:
        ref = O.run_frame(_View(fr, src))
        for p, (a, b) in enumerate(zip(ref, dst)):
            assert_plane_equal(a, b, fr.planes[p]["pixel_type"], "%s plane %d" % (backend, p))


def test_device_generated_frames_match_the_host_generator():
    import torch
    fr = S.SyntheticFrame("YUV422P16LE", 322, 190, seed=0x9F10 + 5)
    for a, b in zip(fr.planes, fr.device_planes(torch.device("cuda", 0))):
        assert np.array_equal(a["src"], b.cpu().numpy())
    fr = S.SyntheticFrame("RGBAF32", 130, 70, seed=77)
    for a, b in zip(fr.planes, fr.device_planes(torch.device("cuda", 0))):
        assert np.array_equal(a["src"], b.cpu().numpy())
    fr = S.SyntheticFrame("NV12", 130, 70, seed=78)
    for a, b in zip(fr.planes, fr.device_planes(torch.device("cuda", 0))):
        assert np.array_equal(a["src"], b.cpu().numpy())


def test_c2_full_size_device_path():
    frames = [S.SyntheticFrame("YUV422P16LE", 3840, 2160, seed=0x9F10 + j, timestamp_ms=1000.0 + 33.3 * j, pixels=False) for j in range(3)]
    check(frames, "yuv_fused_p1")


def test_c2_full_size_device_path_side_stream_lanczos4():
    frames = [S.SyntheticFrame("YUV422P16LE", 3840, 2160, seed=0x9F10 + 40, timestamp_ms=2332.0, interpolation=8, pixels=False)]
    check(frames, "yuv_fused", stream_kind="side")


def test_c3_8k_device_path():
    frames = [S.SyntheticFrame("YUV422P16LE", 7680, 4320, seed=0x9F10 + 9, timestamp_ms=1299.7, pixels=False)]
    check(frames, "yuv_fused_p1")


@pytest.mark.parametrize("fmt", ["RGBAF32", "GBRAPF32LE"])
def test_c4_4k_f32_adaptive_zoom_crop(fmt):
    ov = {"translation2d": (13.25, -7.5)}
    frames = [S.SyntheticFrame(fmt, 3840, 2160, seed=0x9F10 + 21, fov=0.82, base_overrides=ov, pixels=False)]
    check(frames, "yuv_fused")


def test_c1_1080p_nv12_device_path():
    q = S.quat_from_euler_deg(5.0, 2.0, 3.0)
    frames = [S.SyntheticFrame("NV12", 1920, 1080, seed=0x9F10 + j, readout_ms=0.0, constant_quat=q, pixels=False) for j in range(2)]
    check(frames, "yuv_fused")


def test_c2_full_size_address_audit():
    """The audit instantiation of the fused kernel range-checks every tap, store, matrix-row and table address against the
    buffer lengths the caller declared (and re-checks every first-pass certificate): nothing may fall outside on the
    bench workload's geometry — 4K planes that end exactly on a page boundary, device-resident matrices."""
    frames = [S.SyntheticFrame("YUV422P16LE", 3840, 2160, seed=0x9F10 + j, timestamp_ms=1000.0 + 33.3 * j, pixels=False) for j in (0, 17, 63)]
    total = 2 * len(frames) * 3840 * 2160
    for variant, name in ((3, "yuv_fused_p1"),):
        counters = []
        backend, res = run_device_path(frames, variant=variant, audit_out=counters)
        assert backend == name
        certified, wrong, queued, overflow, _, out_of_range = counters[:6]
        assert certified + queued == total and wrong == 0 and overflow == 0
        assert out_of_range == 0
        for fr, src, dst in res:
            ref = O.run_frame(_View(fr, src))
            for p, (a, b) in enumerate(zip(ref, dst)):
                assert_plane_equal(a, b, fr.planes[p]["pixel_type"], "audit plane %d" % p)
