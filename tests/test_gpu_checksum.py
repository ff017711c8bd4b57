"""gfw_set_frame_checksums: a frame's checksum taken where its pixels leave.  The specialised fused kernel adds every element it stores, shifted to its place in the
64-bit word, into a per-lane LDS slot and hands one word per wave to a table that a small kernel adds up behind the launch; every other kernel is followed by a pass
over the region it wrote.  Either way the frame's word must hold what gfw_checksum64 reads from a zero-initialised destination afterwards — and the pixels must be
the oracle's, with and without the option."""
import numpy as np
import pytest

from gyroflow_amd import abi, synthetic as S, warp
import _oracle as O
from test_gpu_parity import assert_plane_equal
from test_gpu_fullsize import _View

pytestmark = pytest.mark.gpu


def run_clip(frames, jit, use_clip, ring=None, variant=0, zero=True, per_plane_calls=False):
    """-> (backend of the last call, [checksum word per ring slot], [gfw_checksum64 of each frame's destination planes], outputs, sources)"""
    import torch
    dev = torch.device("cuda", 0)
    n = len(frames)
    ring = n if ring is None else ring
    d_src = [fr.device_planes(dev) for fr in frames]
    d_dst = [fr.device_outputs(dev) for fr in frames]
    if zero:
        for planes in d_dst:
            for t in planes:
                t.zero_()
    d_mat = [torch.from_numpy(warp.pack_matrices(fr.matrices)).to(dev) for fr in frames]
    d_sums = torch.zeros(ring, dtype=torch.int64, device=dev)
    d_ref = torch.zeros(n, dtype=torch.int64, device=dev)
    torch.cuda.synchronize(dev)
    types = [pl["pixel_type"] for pl in frames[0].planes]
    params = [pl["params"] for pl in frames[0].planes]
    bufs = [[warp.device_buffers(d_src[j][p].data_ptr(), d_src[j][p].numel(), pl["size"], d_dst[j][p].data_ptr(), d_dst[j][p].numel(), pl["out_size"])
             for p, pl in enumerate(fr.planes)] for j, fr in enumerate(frames)]
    rows = frames[0].matrices.shape[0]
    be = warp.Backend(params[0], types[0], frames[0].model, frames[0].digital, bufs[0][0])
    try:
        be.set_stream(torch.cuda.current_stream(dev).cuda_stream)
        be.set_option(abi.OPT_SYNCHRONOUS, 0)
        be.set_option(abi.OPT_MATRICES_ON_DEVICE, 2)
        be.set_option(abi.OPT_JIT, jit)
        if variant:
            be.set_option(abi.OPT_KERNEL_VARIANT, variant)
        be.set_frame_checksums(d_sums.data_ptr(), ring)
        if use_clip:
            warp.ClipCall(be, bufs, params, types, [m.data_ptr() for m in d_mat], rows)()
        else:
            for j in range(n):
                warp.FrameCall(be, bufs[j], params, types, d_mat[j].data_ptr(), rows)()
        be.synchronize()
        backend = warp.last_backend()
        be.set_frame_checksums(0, 0)
        for j in range(n):                                   # the second pass the option replaces: the destination buffers read back
            for t in d_dst[j]:
                assert t.numel() * t.element_size() % 8 == 0
                assert be.lib.gfw_checksum64(be.ctx, t.data_ptr(), t.numel() * t.element_size(), d_ref.data_ptr() + 8 * j) == 0
        be.synchronize()
    finally:
        be.close()
    torch.cuda.synchronize(dev)
    u64 = lambda t: [int(v) & 0xFFFFFFFFFFFFFFFF for v in t.cpu().numpy()]
    return backend, u64(d_sums), u64(d_ref), [[t.cpu().numpy() for t in d_dst[j]] for j in range(n)], [[t.cpu().numpy() for t in d_src[j]] for j in range(n)]


def check_pixels(frames, outs, srcs, what):
    """the pixels are the oracle's (the destinations here start zeroed, the oracle's with the 0x5A fill: compared where a frame writes — its rows, stride padding left out)"""
    for j, fr in enumerate(frames):
        for p, (a, b) in enumerate(zip(O.run_frame(_View(fr, srcs[j])), outs[j])):
            ow, oh, stride = fr.planes[p]["out_size"]
            row_bytes = ow * fr.planes[p]["params"].bytes_per_pixel
            wa, wb = (np.frombuffer(x, np.uint8)[:oh * stride].reshape(oh, stride)[:, :row_bytes] for x in (a, b))
            assert_plane_equal(np.ascontiguousarray(wa).reshape(-1), np.ascontiguousarray(wb).reshape(-1), fr.planes[p]["pixel_type"], "%s, frame %d plane %d" % (what, j, p))


@pytest.mark.parametrize("fmt,kw,n", [
    ("YUV422P16LE", dict(), 19),                          # C2's shape: two full launches and a short one
    ("YUV422P16LE", dict(interpolation=8), 3),
    ("NV12", dict(), 5),
    ("YUV420P", dict(interpolation=4), 3),
    ("P010LE", dict(fov=1.6), 3),
    ("RGBAF32", dict(), 3),
    ("GBRAPF32LE", dict(), 3),
    ("RGBAF16", dict(), 2),
    ("YUV444P16LE", dict(base_overrides={"lens_correction_amount": 0.5}), 2),
])
def test_the_specialised_kernel_takes_each_frames_checksum_in_its_store_path(fmt, kw, n):
    frames = [S.SyntheticFrame(fmt, 320, 192, seed=0xC5 + j, timestamp_ms=1000.0 + 33.3 * j, **kw) for j in range(n)]
    backend, sums, ref, outs, srcs = run_clip(frames, 2, True)
    assert backend.endswith("_jit"), backend
    assert sums == ref and len(set(sums)) == n, (sums, ref)
    check_pixels(frames, outs, srcs, fmt)


def test_frame_by_frame_calls_and_a_ring_shorter_than_the_clip():
    frames = [S.SyntheticFrame("YUV422P16LE", 320, 192, seed=0x15 + j, timestamp_ms=1000.0 + 33.3 * j) for j in range(6)]
    backend, sums, ref, outs, srcs = run_clip(frames, 2, False, ring=4)
    assert backend.endswith("_jit"), backend
    want = [(ref[0] + ref[4]) & 0xFFFFFFFFFFFFFFFF, (ref[1] + ref[5]) & 0xFFFFFFFFFFFFFFFF, ref[2], ref[3]]          # frame k adds to word k mod 4
    assert sums == want, (sums, want)
    check_pixels(frames, outs, srcs, "frame by frame")


@pytest.mark.parametrize("what,jit,variant,kw", [
    ("ahead-of-time fused kernel", 0, 0, dict()),
    ("per-plane kernel", 0, 1, dict()),
    ("EWA sampler (per-plane only)", 2, 0, dict(interpolation=10)),
])
def test_kernels_that_do_not_take_the_sum_are_followed_by_a_pass_over_what_they_wrote(what, jit, variant, kw):
    frames = [S.SyntheticFrame("YUV422P16LE", 322, 190, seed=0x25 + j, timestamp_ms=1000.0 + 33.3 * j, **kw) for j in range(3)]
    backend, sums, ref, outs, srcs = run_clip(frames, jit, True, variant=variant)
    assert not backend.endswith("_jit"), backend
    assert sums == ref and len(set(sums)) == 3, (what, sums, ref)
    check_pixels(frames, outs, srcs, what)


def test_bytes_the_frame_does_not_write_stay_out_of_its_checksum():
    """destinations that are NOT zero to begin with: the frame's word holds the bytes written, so it equals the word of the same frame written over zeros — and the
    full-buffer checksum moves by exactly the bytes that were replaced (nothing to compare it with here: asserted through the zeroed run)."""
    frames = [S.SyntheticFrame("NV12", 322, 190, seed=0x35 + j, timestamp_ms=1000.0 + 33.3 * j) for j in range(2)]
    _, sums_z, ref_z, _, _ = run_clip(frames, 2, True, zero=True)
    _, sums_d, _, _, _ = run_clip(frames, 2, True, zero=False)
    assert sums_z == ref_z and sums_d == sums_z


def test_frames_assembled_from_per_plane_calls_take_their_checksum_on_the_owner_context():
    """The render loop's call sequence (one gfw_undistort_image per plane, each plane a context of its own, coalesced into one fused launch): the frame runs on the
    context that took plane 0 — the option is that context's."""
    import torch
    from test_gpu_coalesce import PlaneLoop, clip
    frames = clip("YUV422P16LE", 320, 192, 5)
    loop = PlaneLoop(frames, jit=2, frames_per_launch=2)
    try:
        dev = loop.dev
        for planes in loop.d_dst:
            for t in planes:
                t.zero_()
        d_sums = torch.zeros(5, dtype=torch.int64, device=dev)
        d_ref = torch.zeros(5, dtype=torch.int64, device=dev)
        torch.cuda.synchronize(dev)
        loop.be[0].set_frame_checksums(d_sums.data_ptr(), 5)
        for j in range(5):
            loop.frame(j)
        loop.be[-1].synchronize()
        names = [warp.Backend.last_backend_of(be) for be in loop.be]
        assert all(n.startswith("yuv_fused") for n in names), names
        loop.be[0].set_frame_checksums(0, 0)
        for j in range(5):
            for t in loop.d_dst[j]:
                assert loop.be[0].lib.gfw_checksum64(loop.be[0].ctx, t.data_ptr(), t.numel(), d_ref.data_ptr() + 8 * j) == 0
        loop.be[0].synchronize()
        torch.cuda.synchronize(dev)
        got, ref = d_sums.cpu().numpy().tolist(), d_ref.cpu().numpy().tolist()
        assert got == ref and len(set(got)) == 5, (got, ref)
    finally:
        loop.close()


@pytest.mark.parametrize("fmt,interp", [("YUV422P16LE", 2), ("NV12", 4), ("RGBAF32", 2)])
def test_output_rects_only_the_rect_is_written_and_only_the_rect_is_summed(fmt, interp):
    """HAS_OUTPUT_RECT / HAS_SOURCE_RECT frames (the per-plane kernel's: plugins' sub-windows): pixels outside the output rect keep the caller's bytes (0x5A here) and
    stay out of the frame's word, which is computed HERE from the oracle's planes at the device addresses of the destinations."""
    import torch
    dev = torch.device("cuda", 0)
    rng = np.random.default_rng(0x4EC7 + interp)
    fr = S.SyntheticFrame(fmt, 322, 190, seed=0x77, fov=0.85, interpolation=interp)
    for pl in fr.planes:
        p = pl["params"]
        (pw, ph), (ow, oh) = pl["size"][:2], pl["out_size"][:2]
        x0, y0 = int(rng.integers(1, pw // 5)), int(rng.integers(1, ph // 5))
        p.source_rect[0], p.source_rect[1], p.source_rect[2], p.source_rect[3] = x0, y0, pw - x0 - int(rng.integers(1, pw // 5)), ph - y0 - int(rng.integers(1, ph // 5))
        x0, y0 = int(rng.integers(1, ow // 5)), int(rng.integers(1, oh // 5))
        p.output_rect[0], p.output_rect[1], p.output_rect[2], p.output_rect[3] = x0, y0, ow - x0 - int(rng.integers(1, ow // 5)), oh - y0 - int(rng.integers(1, oh // 5))
        p.flags |= abi.FLAG_HAS_SOURCE_RECT | abi.FLAG_HAS_OUTPUT_RECT
    d_src, d_dst = fr.device_planes(dev), fr.device_outputs(dev)                   # destinations pre-filled with 0x5A
    d_mat = torch.from_numpy(warp.pack_matrices(fr.matrices)).to(dev)
    d_sum = torch.zeros(1, dtype=torch.int64, device=dev)
    torch.cuda.synchronize(dev)
    types, params = [pl["pixel_type"] for pl in fr.planes], [pl["params"] for pl in fr.planes]
    bufs = [warp.device_buffers(d_src[p].data_ptr(), d_src[p].numel(), pl["size"], d_dst[p].data_ptr(), d_dst[p].numel(), pl["out_size"]) for p, pl in enumerate(fr.planes)]
    be = warp.Backend(params[0], types[0], fr.model, fr.digital, bufs[0])
    try:
        be.set_stream(torch.cuda.current_stream(dev).cuda_stream)
        be.set_option(abi.OPT_SYNCHRONOUS, 0)
        be.set_option(abi.OPT_MATRICES_ON_DEVICE, 2)
        be.set_frame_checksums(d_sum.data_ptr(), 1)
        warp.FrameCall(be, bufs, params, types, d_mat.data_ptr(), fr.matrices.shape[0])()
        be.synchronize()
        assert warp.last_backend() == "plane_generic", warp.last_backend()
    finally:
        be.close()
    torch.cuda.synchronize(dev)
    got = [t.cpu().numpy() for t in d_dst]
    ref = O.run_frame(_View(fr, [t.cpu().numpy() for t in d_src]))
    want = 0
    for p, pl in enumerate(fr.planes):
        assert_plane_equal(ref[p], got[p], pl["pixel_type"], "plane %d" % p)
        assert np.count_nonzero(got[p] == 0x5A) > 0
        q = pl["params"]
        ow, oh, stride = pl["out_size"]
        bpp = q.bytes_per_pixel
        rows = np.frombuffer(ref[p], np.uint8)[:oh * stride].reshape(oh, stride)
        x0, y0, rw, rh = (int(v) for v in q.output_rect)
        sub = rows[y0:y0 + rh, x0 * bpp:(x0 + rw) * bpp]
        pos = (d_dst[p].data_ptr() + (np.arange(y0, y0 + rh, dtype=np.int64)[:, None] * stride) + np.arange(x0 * bpp, (x0 + rw) * bpp, dtype=np.int64)[None, :]) & 7
        for k in range(8):
            want += int(sub[pos == k].astype(np.uint64).sum()) << (8 * k)
    assert (int(d_sum.cpu().numpy()[0]) & 0xFFFFFFFFFFFFFFFF) == (want & 0xFFFFFFFFFFFFFFFF)
