"""Device-side per-row matrix builder (gfw_build_matrices) vs the float64 host statement of
FrameTransform::at_timestamp (frame_transform.rs:221-308).  Tolerance-based by construction: SVD pseudo-inverse and
libm on the host vs closed-form inverse and ocml on the device; the bar is <= 2 ULP of f32 on every entry (relative to
the row's largest entry for values that cancel to ~0)."""
import ctypes as C

import numpy as np
import pytest

from gyroflow_amd import abi, synthetic as S, warp
import _oracle as O

pytestmark = pytest.mark.gpu


def fetch_rows(ptr, rows):
    import torch
    out = torch.empty((rows, 16), dtype=torch.float32, device="cuda")
    # device -> device copy through hipMemcpy exposed by torch: build a tensor view is not possible on a raw pointer,
    # so copy with ctypes' hipMemcpy from the HIP runtime already loaded by torch.
    hip = C.CDLL("libamdhip64.so")
    host = np.empty((rows, 16), dtype=np.float32)
    assert hip.hipMemcpy(C.c_void_p(host.ctypes.data), C.c_void_p(ptr), C.c_size_t(host.nbytes), 2) == 0   # hipMemcpyDeviceToHost
    return host


def ulps(a, b, scale):
    a = a.astype(np.float64); b = b.astype(np.float64)
    ulp = np.spacing(np.maximum(np.abs(a), scale).astype(np.float32)).astype(np.float64)
    return np.abs(a - b) / ulp


@pytest.mark.parametrize("readout_ms,inverted,rot", [(16.0, False, 0.0), (-12.0, False, 0.0), (8.0, True, 0.0), (16.0, False, 90.0), (0.0, False, 0.0)])
def test_device_rows_match_host_f64(readout_ms, inverted, rot):
    w, h = 640, 360
    fr = S.SyntheticFrame("YUV422P16LE", w, h, seed=3)
    org = S.sampled_track(11, 0.0, 2000.0, 1000.0)
    sm = S.sampled_track(12, 0.0, 2000.0, 200.0, scale=0.25)
    nk = S.new_k(fr.lens, 1.0, w, h)
    rows = h if abs(readout_ms) > 0 else 1
    host = S.row_matrices_from_tracks(org, sm, nk, 1000.3, readout_ms, rows, h, rot, inverted, 0.2)
    pl = fr.planes[0]
    b = warp.host_buffers(pl["src"], pl["size"], pl["dst"].copy(), pl["out_size"])
    be = warp.Backend(pl["params"], pl["pixel_type"], fr.model, 0, b)
    try:
        be.set_quaternion_tracks(org, sm)
        ptr = be.build_matrices(nk, 1000.3, readout_ms, rows, h, rot, inverted, 0.2)
        dev = fetch_rows(ptr, rows)
    finally:
        be.close()
    assert np.all(dev[:, 9:14] == 0) and np.all(dev[:, 14] == 1) and np.all(dev[:, 15] == 0)
    scale = np.abs(host[:, :9]).max(axis=1, keepdims=True) * 1e-4          # entries that cancel to ~0 are judged relative to the row
    u = ulps(dev[:, :9], host[:, :9], scale)
    assert u.max() <= 2.0, "max ULP distance %.2f" % u.max()


def test_warp_with_device_built_rows_is_bit_exact_against_oracle_fed_the_same_rows():
    w, h = 640, 360
    fr = S.SyntheticFrame("YUV422P16LE", w, h, seed=9)
    org = S.sampled_track(21, 0.0, 2000.0, 1000.0)
    sm = S.sampled_track(22, 0.0, 2000.0, 200.0, scale=0.25)
    nk = S.new_k(fr.lens, 1.0, w, h)
    outs = [pl["dst"].copy() for pl in fr.planes]
    bufs = [warp.host_buffers(pl["src"], pl["size"], o, pl["out_size"]) for pl, o in zip(fr.planes, outs)]
    params = [pl["params"] for pl in fr.planes]
    types = [pl["pixel_type"] for pl in fr.planes]
    be = warp.Backend(params[0], types[0], fr.model, 0, bufs[0])
    try:
        be.set_quaternion_tracks(org, sm)
        ptr = be.build_matrices(nk, 987.6, 16.0, h, h)
        rows = fetch_rows(ptr, h)
        be.set_option(abi.OPT_MATRICES_ON_DEVICE, 2)
        be.undistort_frame(bufs, params, types, ptr, matrix_count=h)
    finally:
        be.close()
    fr.matrices = np.ascontiguousarray(rows[:, :14])
    ref = O.run_frame(fr)
    for a, b in zip(ref, outs):
        assert np.array_equal(a, b)


def test_async_ring_of_built_tables_overlaps_without_changing_results():
    """Ten frames enqueued back to back in asynchronous mode: each frame's table is built on the context's auxiliary
    stream while earlier frames are still being warped (ring of 4 tables, ordered by events).  Every frame must equal the
    result of the same build + warp done synchronously."""
    import torch
    w, h = 640, 360
    fr = S.SyntheticFrame("YUV422P16LE", w, h, seed=9)
    org = S.sampled_track(21, 0.0, 3000.0, 1000.0)
    sm = S.sampled_track(22, 0.0, 3000.0, 200.0, scale=0.25)
    nk = S.new_k(fr.lens, 1.0, w, h)
    params = [pl["params"] for pl in fr.planes]
    types = [pl["pixel_type"] for pl in fr.planes]
    dev = torch.device("cuda", 0)
    d_src = [torch.from_numpy(pl["src"]).to(dev) for pl in fr.planes]
    stamps = [1000.0 + 33.3 * i for i in range(10)]

    def run(asynchronous):
        d_dst = [[torch.from_numpy(pl["dst"]).to(dev) for pl in fr.planes] for _ in stamps]
        bufs = [[warp.device_buffers(d_src[p].data_ptr(), d_src[p].numel(), pl["size"], d_dst[i][p].data_ptr(), d_dst[i][p].numel(), pl["out_size"])
                 for p, pl in enumerate(fr.planes)] for i in range(len(stamps))]
        be = warp.Backend(params[0], types[0], fr.model, 0, bufs[0][0])
        try:
            be.set_quaternion_tracks(org, sm)
            be.set_option(abi.OPT_MATRICES_ON_DEVICE, 2)
            be.set_option(abi.OPT_SYNCHRONOUS, 0 if asynchronous else 1)
            ptrs = []
            for i, ts in enumerate(stamps):
                ptr = be.build_matrices(nk, ts, 16.0, h, h)
                ptrs.append(ptr)
                be.undistort_frame(bufs[i], params, types, ptr, matrix_count=h)
            be.synchronize()
            assert len(set(ptrs)) == 4                      # the ring
            assert warp.last_backend() == "yuv_fused_p1"
        finally:
            be.close()
        torch.cuda.synchronize()
        return [[t.cpu().numpy() for t in planes] for planes in d_dst]

    sync, asyn = run(False), run(True)
    for i in range(len(stamps)):
        for a, b in zip(sync[i], asyn[i]):
            assert np.array_equal(a, b), "frame %d" % i
    assert not np.array_equal(sync[0][0], sync[5][0])       # the frames do differ


def test_batch_build_equals_single_builds():
    """gfw_build_matrices_batch: every table of a batch is byte-identical to the single-frame build of the same frame."""
    w, h = 640, 360
    fr = S.SyntheticFrame("YUV422P16LE", w, h, seed=9)
    org = S.sampled_track(21, 0.0, 3000.0, 1000.0)
    sm = S.sampled_track(22, 0.0, 3000.0, 200.0, scale=0.25)
    nk = S.new_k(fr.lens, 1.0, w, h)
    pl = fr.planes[0]
    be = warp.Backend(pl["params"], pl["pixel_type"], fr.model, 0, warp.host_buffers(pl["src"], pl["size"], pl["dst"].copy(), pl["out_size"]))
    try:
        be.set_quaternion_tracks(org, sm)
        stamps = [1000.0 + 33.3 * i for i in range(7)]
        ptrs = be.build_matrices_batch(nk, stamps, 16.0, h, h)
        assert len(set(ptrs)) == 7
        batch = [fetch_rows(p, h) for p in ptrs]
        for ts, rows in zip(stamps, batch):
            single = fetch_rows(be.build_matrices(nk, ts, 16.0, h, h), h)
            assert np.array_equal(rows.view(np.uint32), single.view(np.uint32))
        with pytest.raises(warp.GfwError):
            be.build_matrices_batch(nk, [0.0] * 65, 16.0, h, h)
    finally:
        be.close()
