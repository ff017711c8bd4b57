"""The reference's own call sequence — one process_pixels per plane, each plane through its own backend object (src/rendering/mod.rs:494-545) — through
gfw_undistort_image with GFW_OPT_COALESCE_PLANES: asynchronous device-buffer calls are held until the frame's planes have arrived and leave as ONE
launch of the fused frame kernel.  Checked here: the backend that ran (no per-plane kernel), bit-exact planes against the oracle, the completion
contract (gfw_synchronize of ANY member context), host matrices as well as device-resident tables, frames held for a clip launch, and everything
that must break a group apart (a synchronous context, another frame's plane, an error).
"""
import numpy as np
import pytest

from gyroflow_amd import abi, synthetic as S, warp
import _oracle as O
from test_gpu_parity import assert_plane_equal
from test_gpu_fullsize import _View

pytestmark = pytest.mark.gpu


class PlaneLoop:
    """What the render loop holds per plane: a backend object (context) of its own; `frame()` issues the per-plane calls of one frame in order."""

    def __init__(self, frames, device_matrices=True, jit=0, frames_per_launch=1, shared_stream=False, synchronous=False, coalesce=2, own_streams=False, frame_sync=False):
        import torch
        self.torch, self.dev = torch, torch.device("cuda", 0)
        assert abi.load_library().gfw_set_device(0) == 0
        self.frames = frames
        self.d_src = [fr.device_planes(self.dev) for fr in frames]
        self.d_dst = [fr.device_outputs(self.dev) for fr in frames]
        self.d_mat = [torch.from_numpy(warp.pack_matrices(fr.matrices)).to(self.dev) for fr in frames] if device_matrices else None
        torch.cuda.synchronize(self.dev)
        fr0 = frames[0]
        self.bufs = [[warp.device_buffers(self.d_src[j][p].data_ptr(), self.d_src[j][p].numel(), pl["size"], self.d_dst[j][p].data_ptr(), self.d_dst[j][p].numel(), pl["out_size"])
                      for p, pl in enumerate(fr.planes)] for j, fr in enumerate(frames)]
        self.be = []
        stream = torch.cuda.Stream(self.dev) if shared_stream else None
        self.stream = stream
        self.streams = [torch.cuda.Stream(self.dev) for _ in fr0.planes] if own_streams else None      # a caller-owned stream per plane context (the decoder's, say)
        for p, pl in enumerate(fr0.planes):
            be = warp.Backend(pl["params"], pl["pixel_type"], fr0.model, fr0.digital, self.bufs[0][p])
            if stream is not None:
                be.set_stream(stream.cuda_stream)
            if self.streams is not None:
                be.set_stream(self.streams[p].cuda_stream)
            be.set_option(abi.OPT_SYNCHRONOUS, 1 if synchronous else 0)
            be.set_option(abi.OPT_COALESCE_PLANES, coalesce)        # (2: held from the first frame on — what these tests were written against; 1 = the default: test below)
            if frame_sync:
                be.set_option(abi.OPT_FRAME_SYNC, 1)
            if device_matrices:
                be.set_option(abi.OPT_MATRICES_ON_DEVICE, 2)
            be.set_option(abi.OPT_JIT, jit)
            if frames_per_launch > 1:
                be.set_option(abi.OPT_COALESCE_FRAMES, frames_per_launch)
            self.be.append(be)

    def frame(self, j, planes=None):
        fr = self.frames[j]
        for p, pl in enumerate(fr.planes):
            if planes is not None and p not in planes:
                continue
            prm = pl["params"]
            assert prm.plane_index == p
            if self.d_mat is not None:
                self.be[p].undistort_image(self.bufs[j][p], prm, self.d_mat[j].data_ptr(), matrix_count=fr.matrices.shape[0])
            else:
                self.be[p].undistort_image(self.bufs[j][p], prm, fr.matrices)

    def outputs(self, j):
        return [t.cpu().numpy() for t in self.d_dst[j]]

    def check(self, j, what=""):
        fr = self.frames[j]
        ref = O.run_frame(_View(fr, [t.cpu().numpy() for t in self.d_src[j]]))
        for p, (a, b) in enumerate(zip(ref, self.outputs(j))):
            assert_plane_equal(a, b, fr.planes[p]["pixel_type"], "%s frame %d plane %d" % (what, j, p))

    def close(self):
        for be in self.be:
            be.close()


def clip(fmt, w, h, n, **kw):
    return [S.SyntheticFrame(fmt, w, h, seed=0xC0A1 + j, timestamp_ms=1000.0 + 33.3 * j, **kw) for j in range(n)]


@pytest.mark.parametrize("fmt", ["YUV422P16LE", "NV12", "YUV420P", "P010LE", "YUV444P16LE", "GBRAPF32LE"])
@pytest.mark.parametrize("device_matrices", [True, False])
def test_planes_of_a_frame_leave_as_one_fused_launch(fmt, device_matrices):
    loop = PlaneLoop(clip(fmt, 640, 360, 3), device_matrices=device_matrices)
    try:
        for j in range(3):
            loop.frame(j)
            # the last plane's call enqueued the frame: every member context names the fused kernel, none the per-plane one
            names = [warp.Backend.last_backend_of(be) for be in loop.be]
            assert all(n.startswith("yuv_fused") for n in names), names
        loop.be[-1].synchronize()                        # ANY member: its stream is ordered behind the owner's launch
        for j in range(3):
            loop.check(j, fmt)
    finally:
        loop.close()


def test_synchronize_of_a_member_flushes_an_incomplete_frame():
    """Y and U were issued, V never comes: gfw_synchronize of either context must still deliver both planes (per-plane kernel or fused, but complete)."""
    loop = PlaneLoop(clip("YUV422P16LE", 320, 192, 1))
    try:
        loop.frame(0, planes=(0, 1))
        assert warp.Backend.last_backend_of(loop.be[1]) == "held_for_frame"
        loop.be[1].synchronize()
        assert warp.Backend.last_backend_of(loop.be[0]) != "held_for_frame"
        fr = loop.frames[0]
        ref = O.run_frame(_View(fr, [t.cpu().numpy() for t in loop.d_src[0]]))
        out = loop.outputs(0)
        for p in (0, 1):
            assert_plane_equal(ref[p], out[p], fr.planes[p]["pixel_type"], "incomplete frame, plane %d" % p)
        assert np.all(out[2] == 0x5A)                    # nobody asked for V
    finally:
        loop.close()


def test_the_next_frames_first_plane_sends_a_luma_only_frame_on_its_way():
    loop = PlaneLoop(clip("YUV422P16LE", 320, 192, 2))
    try:
        loop.frame(0, planes=(0,))
        loop.frame(1)
        loop.be[0].synchronize()
        fr = loop.frames[0]
        ref = O.run_frame(_View(fr, [t.cpu().numpy() for t in loop.d_src[0]]))
        assert_plane_equal(ref[0], loop.outputs(0)[0], "Luma16", "luma-only frame")
        loop.check(1, "frame behind a luma-only frame")
    finally:
        loop.close()


def test_synchronous_contexts_and_host_buffers_are_never_held():
    loop = PlaneLoop(clip("NV12", 320, 192, 1), synchronous=True)
    try:
        loop.frame(0)
        assert all(warp.Backend.last_backend_of(be) != "held_for_frame" for be in loop.be)
        loop.check(0, "synchronous")                     # no synchronize call: the default contract
    finally:
        loop.close()
    fr = S.SyntheticFrame("NV12", 320, 192, seed=5)
    assert all(np.array_equal(a, b) for a, b in zip(O.run_frame(fr), warp.run_frame(fr, per_plane=True)))


def test_an_invalid_plane_reports_its_own_error_and_releases_the_frame():
    loop = PlaneLoop(clip("YUV422P16LE", 320, 192, 1))
    try:
        loop.frame(0, planes=(0,))
        bad = abi.KernelParams.from_buffer_copy(loop.frames[0].planes[1]["params"])
        bad.interpolation = 7
        with pytest.raises(Exception):
            loop.be[1].undistort_image(loop.bufs[0][1], bad, loop.d_mat[0].data_ptr(), matrix_count=loop.frames[0].matrices.shape[0])
        assert warp.Backend.last_backend_of(loop.be[0]) != "held_for_frame"      # plane 0 went out on its own
        loop.be[0].synchronize()
        fr = loop.frames[0]
        ref = O.run_frame(_View(fr, [t.cpu().numpy() for t in loop.d_src[0]]))
        assert_plane_equal(ref[0], loop.outputs(0)[0], "Luma16", "plane 0 after plane 1 failed")
    finally:
        loop.close()


@pytest.mark.parametrize("shared_stream", [False, True])
def test_frames_held_for_a_clip_launch_of_the_specialised_kernel(shared_stream):
    """GFW_OPT_COALESCE_FRAMES = 4: the per-plane calls of four frames become ONE launch of the run-time specialised kernel (what gfw_undistort_clip does
    for callers that own the frame loop); a member's gfw_synchronize delivers frames the owner still holds."""
    n = 10
    loop = PlaneLoop(clip("YUV422P16LE", 384, 208, n), jit=2, frames_per_launch=4, shared_stream=shared_stream)
    try:
        for j in range(n):
            loop.frame(j)
        loop.be[2].synchronize()                         # frames 8, 9 are still held by the owner (context of plane 0): a member's synchronise sends them
        assert warp.Backend.last_backend_of(loop.be[0]).endswith("_jit")
        ms, launches, frames = (0.0, 0, 0)
        for j in range(n):
            loop.check(j, "clip launch")
    finally:
        loop.close()


def test_option_off_restores_the_per_plane_launches():
    loop = PlaneLoop(clip("YUV422P16LE", 320, 192, 1))
    try:
        for be in loop.be:
            be.set_option(abi.OPT_COALESCE_PLANES, 0)
        loop.frame(0)
        names = [warp.Backend.last_backend_of(be) for be in loop.be]
        assert "held_for_frame" not in names and names[1] == names[2] == "plane_generic", names      # (a lone luma plane is a frame the fused kernel serves)
        for be in loop.be:
            be.synchronize()
        loop.check(0, "coalescing off")
    finally:
        loop.close()


def test_default_holds_only_contexts_seen_as_planes_of_a_multi_plane_frame():
    """GFW_OPT_COALESCE_PLANES = 1 (the default): the first frame of a clip leaves plane by plane — nothing has shown yet that these contexts are planes of one
    frame — and marks them; from the second frame on the calls are held and leave fused.  A lone luma context (greyscale) is never held: its call launches."""
    loop = PlaneLoop(clip("YUV422P16LE", 320, 192, 3), coalesce=1)
    try:
        loop.frame(0)
        names = [warp.Backend.last_backend_of(be) for be in loop.be]
        assert "held_for_frame" not in names and names[1] == names[2] == "plane_generic", names
        for j in (1, 2):
            loop.frame(j, planes=(0, 1))
            assert warp.Backend.last_backend_of(loop.be[1]) == "held_for_frame"
            loop.frame(j, planes=(2,))
            names = [warp.Backend.last_backend_of(be) for be in loop.be]
            assert all(n.startswith("yuv_fused") for n in names), names
        loop.be[1].synchronize()
        for j in range(3):
            loop.check(j, "default option")
    finally:
        loop.close()
    grey = PlaneLoop([S.SyntheticFrame("YUV422P16LE", 320, 192, seed=0xC0A1 + j) for j in range(3)], coalesce=1)
    try:
        for j in range(3):
            grey.frame(j, planes=(0,))                   # one context, plane_index 0, again and again: never a multi-plane pattern
            assert warp.Backend.last_backend_of(grey.be[0]) != "held_for_frame"
        import torch
        torch.cuda.current_stream().synchronize()
        torch.cuda.synchronize()                         # raw device synchronisation, no gfw call: the planes must be there
        for j in range(3):
            fr = grey.frames[j]
            ref = O.run_frame(_View(fr, [t.cpu().numpy() for t in grey.d_src[j]]))
            assert_plane_equal(ref[0], grey.outputs(j)[0], "Luma16", "greyscale frame %d" % j)
    finally:
        grey.close()


def test_the_fused_launch_waits_for_work_queued_on_a_member_contexts_stream():
    """Input-side ordering: each plane context has its caller's stream; the U plane's pixels are UPLOADED on the U context's stream right before
    its call, behind a long-running kernel.  The fused launch runs on the Y context's stream: it must wait for the U stream's pending work, else it reads the
    plane before the upload lands."""
    import torch
    loop = PlaneLoop(clip("YUV422P16LE", 640, 360, 2), own_streams=True)
    try:
        for j in range(2):
            good = [t.clone() for t in loop.d_src[j]]
            for p in (1, 2):
                loop.d_src[j][p].fill_(0x11)             # stale content
            torch.cuda.synchronize()
            for p in (1, 2):
                with torch.cuda.stream(loop.streams[p]):
                    torch.cuda._sleep(40_000_000)        # ~20 ms of GPU time ahead of the upload on this plane's stream
                    loop.d_src[j][p].copy_(good[p], non_blocking=True)
            loop.frame(j)
            loop.be[0].synchronize()
            torch.cuda.synchronize()
            loop.check(j, "member-stream upload")
    finally:
        loop.close()


def test_frames_with_their_own_parameters_do_not_share_a_clip_launch():
    """GFW_OPT_COALESCE_FRAMES > 1 with per-call parameters that move from frame to frame (dynamic zoom: fov; a keyframed lens correction is the same
    mechanism): a launch shares ONE parameter block, so a frame whose block differs must not join it — every frame matches the oracle fed
    ITS parameters."""
    n = 6
    frames = [S.SyntheticFrame("YUV422P16LE", 384, 208, seed=0xC0A1 + j, timestamp_ms=1000.0 + 33.3 * j, fov=1.0 + 0.05 * (j // 2), base_overrides={"lens_correction_amount": 0.8}) for j in range(n)]
    loop = PlaneLoop(frames, jit=2, frames_per_launch=4)
    try:
        for j in range(n):
            loop.frame(j)
        loop.be[1].synchronize()
        for j in range(n):
            loop.check(j, "per-frame fov")
    finally:
        loop.close()


def test_frame_sync_holds_the_planes_of_synchronous_contexts_until_the_last():
    """GFW_OPT_FRAME_SYNC: synchronous contexts (the reference's contract) whose caller consumes a frame only after its last plane's call: the earlier planes
    are held, the last plane's call launches the fused kernel and returns when the frame is complete — no synchronize call afterwards."""
    loop = PlaneLoop(clip("YUV422P16LE", 640, 360, 3), synchronous=True, frame_sync=True)
    try:
        for j in range(3):
            loop.frame(j, planes=(0, 1))
            assert warp.Backend.last_backend_of(loop.be[1]) == "held_for_frame"
            loop.frame(j, planes=(2,))
            names = [warp.Backend.last_backend_of(be) for be in loop.be]
            assert all(n.startswith("yuv_fused") for n in names), names
            loop.check(j, "frame sync")                  # straight after the last plane's call
    finally:
        loop.close()


def test_get_stream_sends_held_planes_on_their_way():
    """A consumer that orders itself behind the context's stream (an event, a synchronise of the raw stream) asks for the stream first: gfw_get_stream flushes."""
    import torch
    loop = PlaneLoop(clip("YUV422P16LE", 320, 192, 1))
    try:
        loop.frame(0, planes=(0, 1))
        assert warp.Backend.last_backend_of(loop.be[1]) == "held_for_frame"
        assert loop.be[1].get_stream() is not None
        assert warp.Backend.last_backend_of(loop.be[1]) != "held_for_frame"
        torch.cuda.synchronize()
        fr = loop.frames[0]
        ref = O.run_frame(_View(fr, [t.cpu().numpy() for t in loop.d_src[0]]))
        out = loop.outputs(0)
        for p in (0, 1):
            assert_plane_equal(ref[p], out[p], fr.planes[p]["pixel_type"], "after gfw_get_stream, plane %d" % p)
    finally:
        loop.close()


def test_threads_assembling_frames_side_by_side_do_not_wait_for_each_others_launches():
    """Round 5: the process-wide lock covers the book-keeping of held planes, not the launches.  Four render threads, each with the plane contexts
    of its own clip — one of them behind a blocking run-time build (GFW_OPT_JIT = 2, ~1 s) — assemble their frames concurrently; every frame is bit-exact,
    and the threads without a build are done long before the one with it (they used to queue behind the lock it held)."""
    import threading, time
    fmts = ["YUV422P16LE", "NV12", "YUV420P", "P010LE"]
    loops = [PlaneLoop(clip(f, 640, 360, 4), jit=2 if i == 0 else 0, frames_per_launch=2 if i == 3 else 1) for i, f in enumerate(fmts)]
    done, errors = [None] * len(loops), []
    gate = threading.Barrier(len(loops))

    def work(i):
        try:
            gate.wait()
            for j in range(4):
                loops[i].frame(j)
            loops[i].be[-1].synchronize()
            done[i] = time.perf_counter()
        except Exception as e:                     # noqa: BLE001 — reported below, in the main thread
            errors.append((i, repr(e)))

    try:
        ts = [threading.Thread(target=work, args=(i,)) for i in range(len(loops))]
        t0 = time.perf_counter()
        for t in ts:
            t.start()
        for t in ts:
            t.join(120)
        assert not errors, errors
        assert all(d is not None for d in done)
        for i, loop in enumerate(loops):
            for j in range(4):
                loop.check(j, "thread %d %s" % (i, fmts[i]))
        build = done[0] - t0
        if build > 0.3:                            # (a cached code object makes the build instantaneous: nothing to compare then)
            assert max(done[1:]) - t0 < 0.8 * build, [d - t0 for d in done]      # (behind the lock they would end AFTER the build; measured: 0.02-0.05 s against ~1 s)
    finally:
        for loop in loops:
            loop.close()
