"""Decoder surfaces without the host: gfw_import_external_fd maps an allocation ANOTHER PROCESS exported as a file descriptor —
the MI355X counterpart of the reference's cuImportExternalMemory -> cuExternalMemoryGetMappedBuffer (src/core/gpu/wgpu_interop_cuda.rs:181-215) — and the
warp reads its source planes straight from that mapping: the pixels never exist in this process's host memory on the device side of the call.
tests/cpp/export_surface (built by __graft_entry__.build) plays the decoder."""
import ctypes as C
import os
import socket
import subprocess
import tempfile

import numpy as np
import pytest

from gyroflow_amd import abi, synthetic as S, warp
import _oracle as O
from test_gpu_parity import assert_plane_equal

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
EXPORTER = os.path.join(HERE, "cpp", "export_surface")


def export_from_another_process(blob, tmp):
    """-> (fd, size, finish()): the helper process owns the allocation until finish() is called"""
    if not os.path.exists(EXPORTER):
        pytest.skip("tests/cpp/export_surface not built (python __graft_entry__.py)")
    path = os.path.join(tmp, "s.sock")
    data = os.path.join(tmp, "surface.bin")
    open(data, "wb").write(blob)
    srv = socket.socket(socket.AF_UNIX, socket.SOCK_STREAM)
    srv.bind(path); srv.listen(1); srv.settimeout(60)
    proc = subprocess.Popen([EXPORTER, path, data], stderr=subprocess.PIPE)
    try:
        conn, _ = srv.accept()
    except socket.timeout:
        proc.kill()
        pytest.fail("exporter did not connect: %s" % proc.stderr.read().decode()[-400:])
    msg, fds, _, _ = socket.recv_fds(conn, 8, 1)
    assert len(fds) == 1, (msg, fds, proc.poll())
    size = int.from_bytes(msg, "little")

    def finish():
        try:
            conn.send(b"x")
        finally:
            conn.close(); srv.close()
            assert proc.wait(timeout=30) == 0
    return fds[0], size, finish


@pytest.mark.parametrize("fmt", ["YUV422P16LE", "NV12"])
def test_warp_reads_a_surface_exported_by_another_process(fmt):
    import torch
    dev = torch.device("cuda", 0)
    lib = abi.load_library()
    assert lib.gfw_set_device(0) == 0
    fr = S.SyntheticFrame(fmt, 1280, 720, seed=0x1D0)
    # the decoder's surface: the planes one after the other at 256-byte aligned offsets, pitch = the frame's stride
    offs, blob = [], bytearray()
    for pl in fr.planes:
        while len(blob) % 256:
            blob.append(0)
        offs.append(len(blob)); blob += pl["src"].tobytes()
    with tempfile.TemporaryDirectory() as tmp:
        fd, size, finish = export_from_another_process(bytes(blob), tmp)
        ptr, handle = C.c_void_p(0), C.c_void_p(0)
        try:
            # a tiled surface is refused, whatever the descriptor
            assert lib.gfw_import_external_fd(fd, size, 0x0200000000000001, C.byref(ptr), C.byref(handle)) == -7
            rc = lib.gfw_import_external_fd(fd, size, 0, C.byref(ptr), C.byref(handle))
            assert rc == 0 and ptr.value, lib.gfw_last_error().decode()
            d_dst = fr.device_outputs(dev)
            torch.cuda.synchronize(dev)                      # (torch fills the destinations on ITS stream; the context's stream is non-blocking: without this the 0x5A fill can land after the warp — seen under pytest -n 4, r06_b)
            bufs = [warp.device_buffers(ptr.value + offs[p], pl["src"].nbytes, pl["size"], d_dst[p].data_ptr(), d_dst[p].numel(), pl["out_size"]) for p, pl in enumerate(fr.planes)]
            be = warp.Backend(fr.planes[0]["params"], fr.planes[0]["pixel_type"], fr.model, fr.digital, bufs[0])
            try:
                be.set_option(abi.OPT_SYNCHRONOUS, 0)
                be.undistort_frame(bufs, [pl["params"] for pl in fr.planes], [pl["pixel_type"] for pl in fr.planes], fr.matrices)
                be.synchronize()
                assert warp.last_backend().startswith("yuv_fused")
            finally:
                be.close()
            for p, (a, b) in enumerate(zip(O.run_frame(fr), d_dst)):
                assert_plane_equal(a, b.cpu().numpy(), fr.planes[p]["pixel_type"], "imported surface, plane %d" % p)
        finally:
            if handle.value:
                assert lib.gfw_release_external(handle) == 0
            os.close(fd)
            finish()


def test_import_rejects_what_is_not_a_device_allocation():
    lib = abi.load_library()
    ptr, handle = C.c_void_p(0), C.c_void_p(0)
    assert lib.gfw_import_external_fd(-1, 4096, 0, C.byref(ptr), C.byref(handle)) == -9
    with tempfile.TemporaryFile() as f:                      # a regular file is not a dma-buf
        f.write(b"\0" * 4096); f.flush()
        rc = lib.gfw_import_external_fd(f.fileno(), 4096, 0, C.byref(ptr), C.byref(handle))
        assert rc == -7 and not handle.value and "not a dma-buf" in lib.gfw_last_error().decode(), (rc, lib.gfw_last_error())
