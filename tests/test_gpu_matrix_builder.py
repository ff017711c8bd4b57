"""Device-side per-row matrix builder (gfw_build_matrices) vs the float64 host statement of
FrameTransform::at_timestamp (frame_transform.rs:221-308).  Tolerance-based by construction: SVD pseudo-inverse and
libm on the host vs closed-form inverse and ocml on the device; the bar is <= 2 ULP of f32 on every entry (relative to
the row's largest entry for values that cancel to ~0)."""
import ctypes as C

import numpy as np
import pytest

from gyroflow_amd import abi, synthetic as S, warp
import _oracle as O
import _hoststmt as HS

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
    host = HS.row_matrices_from_tracks(org, sm, nk, 1000.3, readout_ms, rows, h, rot, inverted, 0.2)
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
            assert warp.last_backend().startswith("yuv_fused_p1")
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


def _rows(fr, fn):
    pl = fr.planes[0]
    b = warp.host_buffers(pl["src"], pl["size"], pl["dst"].copy(), pl["out_size"])
    be = warp.Backend(pl["params"], pl["pixel_type"], fr.model, 0, b)
    try:
        return fn(be)
    finally:
        be.close()


def test_sync_offsets_shift_every_row_lookup_and_zero_duration_gives_identity():
    """gyro_source/mod.rs:857-860: `timestamp_ms -= offset_at_video_timestamp(timestamp_ms)` per lookup (so per row), identity
    quaternions when duration_ms <= 0."""
    w, h = 640, 360
    fr = S.SyntheticFrame("YUV422P16LE", w, h, seed=3)
    org = S.sampled_track(11, 0.0, 2500.0, 1000.0)
    sm = S.sampled_track(12, 0.0, 2500.0, 200.0, scale=0.25)
    nk = S.new_k(fr.lens, 1.0, w, h)
    offs = (np.array([200000, 900000, 1004000, 1700000], dtype=np.int64), np.array([12.5, -30.25, 41.0, 7.75]))   # us -> ms
    one = (np.array([500000], dtype=np.int64), np.array([-17.5]))
    for offsets in (offs, one):
        host = HS.row_matrices_from_tracks(org, sm, nk, 1000.3, 16.0, h, h, offsets=offsets, duration_ms=2500.0)
        plain = HS.row_matrices_from_tracks(org, sm, nk, 1000.3, 16.0, h, h)
        assert not np.array_equal(host, plain)

        def run(be):
            be.set_quaternion_tracks(org, sm)
            be.set_sync_offsets(2500.0, offsets[0], offsets[1])
            return fetch_rows(be.build_matrices(nk, 1000.3, 16.0, h, h), h)
        dev = _rows(fr, run)
        scale = np.abs(host[:, :9]).max(axis=1, keepdims=True) * 1e-4
        assert ulps(dev[:, :9], host[:, :9], scale).max() <= 2.0

    def run0(be):
        be.set_quaternion_tracks(org, sm)
        be.set_sync_offsets(0.0)
        return fetch_rows(be.build_matrices(nk, 1000.3, 16.0, h, h), h)
    dev = _rows(fr, run0)
    host = HS.row_matrices_from_tracks(org, sm, nk, 1000.3, 16.0, h, h, duration_ms=0.0)
    assert np.all(dev[:, :9] == dev[0, :9])                     # identity rotation on every row
    scale = np.abs(host[:, :9]).max(axis=1, keepdims=True) * 1e-4
    assert ulps(dev[:, :9], host[:, :9], scale).max() <= 2.0


def _stab(w, h):
    pos = np.linspace(-200.0, 3400.0, 19)
    ibis = np.stack([pos, 14.0 * np.sin(pos * 0.004), -9.0 * np.cos(pos * 0.003), 350.0 * np.sin(pos * 0.002 + 0.4)], axis=1)
    ois = np.stack([pos, 3.0 * np.cos(pos * 0.005), 2.0 * np.sin(pos * 0.006), np.zeros_like(pos)], axis=1)
    return {"offset": 12.5, "sensor_size": (6000.0, 3376.0), "crop_area": (120.0, 338.0, 5760.0, 2700.0), "pixel_pitch": (3.0, 3.0),
            "width": float(w), "height": float(h), "ibis": ibis, "ois": ois}


@pytest.mark.parametrize("inverted", [False, True])
def test_ibis_ois_spline_terms_per_row(inverted):
    """frame_transform.rs:234-241, :270-289: Catmull-Rom evaluation of the stabiliser positions at each row's sensor line;
    the f32 terms must equal the host statement's bit for bit or within 1 ULP (f64 evaluation, one rounding), the roll's cos/sin
    slots must be the host libm's of that very f32 angle."""
    w, h = 640, 360
    fr = S.SyntheticFrame("YUV422P16LE", w, h, seed=3)
    org = S.sampled_track(11, 0.0, 2000.0, 1000.0)
    sm = S.sampled_track(12, 0.0, 2000.0, 200.0, scale=0.25)
    nk = S.new_k(fr.lens, 1.0, w, h)
    stab = _stab(w, h)
    host = HS.row_matrices_from_tracks(org, sm, nk, 1000.3, 16.0, h, h, framebuffer_inverted=inverted, stab=stab)

    def run(be):
        be.set_quaternion_tracks(org, sm)
        return fetch_rows(be.build_matrices(nk, 1000.3, 16.0, h, h, framebuffer_inverted=inverted, stab=stab), h)
    dev = _rows(fr, run)
    assert np.abs(host[:, 9:14]).max() > 0.5                     # the terms are really there
    t = ulps(dev[:, 9:14], host[:, 9:14], np.full((h, 1), 1e-6))
    assert t.max() <= 1.0, "IBIS/OIS terms differ by %.2f ULP" % t.max()
    with np.errstate(all="ignore"):
        want_cos = np.cos((-dev[:, 11]).astype(np.float32), dtype=np.float32)
    # cos/sin slots: exactly what the host libm returns for the f32 angle the row carries (what cpu_undistort.rs:159-160 evaluates)
    lib = O.lib()
    ang = np.ascontiguousarray(-dev[:, 11], dtype=np.float32)
    c = np.empty_like(ang); sn = np.empty_like(ang)
    lib.gfw_oracle_libm(3, ang.ctypes.data, c.ctypes.data, ang.size)
    lib.gfw_oracle_libm(2, ang.ctypes.data, sn.ctypes.data, ang.size)
    assert np.array_equal(dev[:, 14].view(np.uint32), c.view(np.uint32)) and np.array_equal(dev[:, 15].view(np.uint32), sn.view(np.uint32))
    scale = np.abs(host[:, :9]).max(axis=1, keepdims=True) * 1e-4
    assert ulps(dev[:, :9], host[:, :9], scale).max() <= 2.0


@pytest.mark.parametrize("mode", [1, 2])
def test_suppress_rotation(mode):
    w, h = 640, 360
    fr = S.SyntheticFrame("YUV422P16LE", w, h, seed=3)
    org = S.sampled_track(11, 0.0, 2000.0, 1000.0)
    sm = S.sampled_track(12, 0.0, 2000.0, 200.0, scale=0.25)
    nk = S.new_k(fr.lens, 1.0, w, h)
    stab = _stab(w, h)
    host = HS.row_matrices_from_tracks(org, sm, nk, 1000.3, 16.0, h, h, suppress_rotation=mode, stab=stab)

    def run(be):
        be.set_quaternion_tracks(org, sm)
        return fetch_rows(be.build_matrices(nk, 1000.3, 16.0, h, h, suppress_rotation=mode, stab=stab), h)
    dev = _rows(fr, run)
    assert np.all(dev[:, :9] == dev[0, :9])                     # R = identity on every row
    scale = np.abs(host[:, :9]).max(axis=1, keepdims=True) * 1e-4
    assert ulps(dev[:, :9], host[:, :9], scale).max() <= 2.0
    if mode == 2:
        assert np.all(dev[:, 9:14] == 0) and np.all(dev[:, 14] == 1) and np.all(dev[:, 15] == 0)
    else:
        assert ulps(dev[:, 9:14], host[:, 9:14], np.full((h, 1), 1e-6)).max() <= 1.0


def test_suppress_rotation_outside_its_values_is_rejected():
    # the slot was padding until ABI version 1: a caller that leaves it uninitialised must hear about it, not get R = identity
    w, h = 320, 192
    fr = S.SyntheticFrame("YUV422P16LE", w, h, seed=3)
    nk = S.new_k(fr.lens, 1.0, w, h)

    def run(be):
        be.set_quaternion_tracks(S.sampled_track(11, 0.0, 2000.0, 1000.0), S.sampled_track(12, 0.0, 2000.0, 200.0, scale=0.25))
        for bad in (3, -1, 0x5A5A5A5A):
            with pytest.raises(warp.GfwError) as e:
                be.build_matrices(nk, 1000.3, 16.0, h, h, suppress_rotation=bad)
            assert e.value.code == abi.ERR_INVALID_ARGUMENT and "suppress_rotation" in str(e.value)
        return None
    _rows(fr, run)


def test_stabiliser_builds_stay_asynchronous_and_ordered():
    """Six consecutive frames with different IBIS/OIS control points on one context: each table must carry ITS frame's terms (the
    control points travel through a ring of pinned / device buffers on the building stream, no stream-wide synchronisation)."""
    w, h = 640, 360
    fr = S.SyntheticFrame("YUV422P16LE", w, h, seed=3)
    org = S.sampled_track(11, 0.0, 2000.0, 1000.0)
    sm = S.sampled_track(12, 0.0, 2000.0, 200.0, scale=0.25)
    nk = S.new_k(fr.lens, 1.0, w, h)
    stabs = []
    for j in range(6):
        st = _stab(w, h)
        st["ibis"] = np.asarray(st["ibis"], dtype=np.float64).copy()
        st["ibis"][:, 1:] *= (1.0 + 0.37 * j)
        stabs.append(st)

    def run(be):
        be.set_quaternion_tracks(org, sm)
        be.set_option(abi.OPT_SYNCHRONOUS, 0)
        ptrs = [be.build_matrices(nk, 1000.3 + 33.3 * j, 16.0, h, h, stab=stabs[j]) for j in range(4)]      # the ring holds four tables
        be.synchronize()
        return [fetch_rows(p, h) for p in ptrs]
    devs = _rows(fr, run)
    for j, dev in enumerate(devs):
        host = HS.row_matrices_from_tracks(org, sm, nk, 1000.3 + 33.3 * j, 16.0, h, h, stab=stabs[j])
        assert ulps(dev[:, 9:14], host[:, 9:14], np.full((h, 1), 1e-6)).max() <= 1.0, j


def test_warp_with_device_built_ibis_rows_is_bit_exact_against_the_oracle():
    w, h = 640, 360
    fr = S.SyntheticFrame("YUV422P16LE", w, h, seed=9, flags=abi.FLAG_HAS_IBIS_DATA)
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
        ptr = be.build_matrices(nk, 987.6, 16.0, h, h, stab=_stab(w, h))
        rows = fetch_rows(ptr, h)
        be.set_option(abi.OPT_MATRICES_ON_DEVICE, 2)
        be.undistort_frame(bufs, params, types, ptr, matrix_count=h)
        assert warp.last_backend() == "yuv_fused"              # IBIS terms: the generic-model instantiation
    finally:
        be.close()
    fr.matrices = np.ascontiguousarray(rows[:, :14])
    assert np.abs(fr.matrices[:, 9:14]).max() > 0.5
    ref = O.run_frame(fr)
    for a, b in zip(ref, outs):
        assert np.array_equal(a, b)
