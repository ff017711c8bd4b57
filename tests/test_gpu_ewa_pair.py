"""EWA on planar chroma (round 6): the U and V planes of a planar frame share every kernel parameter but plane_index and the background, so their coordinates,
jacobians and tap weights are the same numbers — the library warps both in ONE launch of the per-plane kernel (gfw_plane_kernel<.., DUAL>, gfw_api.hip run_planes):
each plane must still come out as the oracle writes it on its own (cpu_undistort.rs:331-369 per plane), with its own background, and planes that do not share their
parameters must not be paired."""
import ctypes as C

import numpy as np
import pytest

from gyroflow_amd import abi, synthetic as S, warp
import _oracle as O
from test_gpu_parity import assert_plane_equal

pytestmark = pytest.mark.gpu


def run_counted(fr, mutate=None):
    """the frame through gfw_undistort_frame from HOST buffers on a context of its own -> (outputs, backend, launches that served two planes)"""
    outs = [pl["dst"].copy() for pl in fr.planes]
    bufs = [warp.host_buffers(pl["src"], pl["size"], o, pl["out_size"]) for pl, o in zip(fr.planes, outs)]
    params = [pl["params"] for pl in fr.planes]
    types = [pl["pixel_type"] for pl in fr.planes]
    lib = abi.load_library()
    lib.gfw_debug_paired_launches.argtypes = [C.c_void_p]
    lib.gfw_debug_paired_launches.restype = C.c_longlong
    be = warp.Backend(params[0], types[0], fr.model, fr.digital, bufs[0])
    try:
        be.undistort_frame(bufs, params, types, fr.matrices)
        paired = int(lib.gfw_debug_paired_launches(be.ctx))
        backend = warp.last_backend()
    finally:
        be.close()
    return outs, backend, paired


@pytest.mark.parametrize("fmt,interp,kw,want", [
    ("YUV422P16LE", 10, dict(fov=1.6, background_rgba=(0.9, 0.2, 0.4, 1.0)), 1),
    ("YUV420P", 12, dict(fov=1.4, background_rgba=(0.1, 0.8, 0.3, 1.0), base_overrides={"background_mode": 1}), 1),
    ("YUV444P16LE", 11, dict(fov=1.3, background_rgba=(0.3, 0.6, 0.9, 1.0), base_overrides={"background_mode": 3, "background_margin": 0.1, "background_margin_feather": 0.12}), 1),
    ("YUV420P", 13, dict(fov=1.2, flags=abi.FLAG_FIX_COLOR_RANGE, limited_range=True), 1),
    ("YUV420P10LE", 10, dict(fov=0.9), 1),
    ("GBRAPF32LE", 10, dict(fov=1.4, background_rgba=(0.7, 0.1, 0.5, 0.9)), 2),        # four float planes of one geometry, no colour-range fix: 0 + 1 and 2 + 3
    ("YUVA444P10LE", 12, dict(fov=1.3, background_rgba=(0.2, 0.9, 0.4, 0.6)), 2),      # 4:4:4 with alpha, no colour-range fix: Y + U, V + A
    ("NV12", 10, dict(fov=1.5, background_rgba=(0.9, 0.2, 0.4, 1.0)), 0),              # interleaved chroma is one two-channel plane already
    ("RGBA", 12, dict(fov=1.5), 0),
    ("YUV422P16LE", 4, dict(fov=1.5), 0),                                               # not EWA: the fused kernel's frame
])
def test_u_and_v_in_one_launch_equal_the_oracle_plane_by_plane(fmt, interp, kw, want):
    fr = S.SyntheticFrame(fmt, 322, 190, seed=0xE3A + interp, interpolation=interp, **kw)
    ref = O.run_frame(fr)
    got, backend, paired = run_counted(fr)
    assert paired == want, (backend, paired)
    assert (backend == "plane_generic") == (interp >= 10), backend
    for i, (a, b) in enumerate(zip(ref, got)):
        assert_plane_equal(a, b, fr.planes[i]["pixel_type"], "%s EWA %d plane %d" % (fmt, interp, i))


def test_planes_that_do_not_share_their_parameters_are_not_paired():
    """V with another pixel_value_limit than U: the launch of U cannot stand for it (and each plane still equals the oracle)."""
    fr = S.SyntheticFrame("YUV422P16LE", 322, 190, seed=0xE3B, interpolation=10, fov=1.4, background_rgba=(0.5, 0.1, 0.7, 1.0))
    fr.planes[2]["params"].pixel_value_limit = 40000.0
    ref = O.run_frame(fr)
    got, backend, paired = run_counted(fr)
    assert backend == "plane_generic" and paired == 0, (backend, paired)
    for i, (a, b) in enumerate(zip(ref, got)):
        assert_plane_equal(a, b, fr.planes[i]["pixel_type"], "unpaired plane %d" % i)
    assert np.max(np.frombuffer(got[2].tobytes(), dtype=np.uint16)) <= 40000
