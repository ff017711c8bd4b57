"""Runs LAST in a serial run of the GPU tier (the driver's `pytest tests -x -q -m gpu`: one process): after everything the suite has specialised, one more clip must
still get its run-time specialised kernel.  Round 6's suite outgrew the 256 specialisations a process used to keep — the clips after that were served ahead of time
("specialisation cache full": correct pixels, wrong kernel) and tests that name the specialised kernel failed two thirds of the way through the suite
(gfw_jit.hip kMaxEntries, now 4096)."""
import pytest

from gyroflow_amd import synthetic as S, warp
import _oracle as O
from test_gpu_parity import assert_plane_equal

pytestmark = pytest.mark.gpu


def test_the_process_can_still_specialise_a_clip_after_the_whole_suite():
    fr = S.SyntheticFrame("YUV422P16LE", 322, 190, seed=0x2A2A, fov=1.234567)           # constants no other test uses: a specialisation of its own
    ref = O.run_frame(fr)
    got = warp.run_frame(fr, jit=2)
    assert warp.last_backend().endswith("_jit"), warp.last_backend()
    for i, (a, b) in enumerate(zip(ref, got)):
        assert_plane_equal(a, b, fr.planes[i]["pixel_type"], "plane %d" % i)
