"""The GPU tier's seeded random sweep (tests/test_gpu_fuzz.py: format x interpolation x background mode x flags x lens model x digital lens x readout
direction x odd sizes x blend x stretch x margins, every other of its 80 configurations; 1 300 more went through offline: profiles/r03_interpreter_hunts.txt) through the host-interpreted kernel sources (tests/_emu.py): the per-plane
kernel on every configuration, the fused kernel wherever the library would use it — both bit-identical to the oracle."""
import numpy as np
import pytest

from gyroflow_amd import synthetic as S
import _emu
import _oracle as O
from test_gpu_fuzz import random_case


@pytest.mark.parametrize("seed", range(0, 80, 2))
def test_random_configuration_through_the_interpreted_kernels(seed):
    fmt, w, h, kw = random_case(seed)
    fr = S.SyntheticFrame(fmt, w, h, **kw)
    ref = O.run_frame(fr)
    what = "seed %d %s %dx%d %s" % (seed, fmt, w, h, kw)
    for i, (a, b) in enumerate(zip(ref, _emu.run_frame_per_plane(fr))):
        assert np.array_equal(a, b), "per-plane kernel, plane %d: %d bytes differ (%s)" % (i, int(np.count_nonzero(a != b)), what)
    if _emu.fused_eligible(fr):
        for i, (a, b) in enumerate(zip(ref, _emu.run_frame(fr))):
            assert np.array_equal(a, b), "fused kernel, plane %d: %d bytes differ (%s)" % (i, int(np.count_nonzero(a != b)), what)


@pytest.mark.parametrize("seed", range(1, 80, 8))
def test_random_configuration_through_the_checksum_build(seed):
    """gfw_set_frame_checksums: the same generator through the checksum build of the fused kernel (GFW_BK_checksum) — the frame's word must be the checksum of the
    bytes written, whichever store path wrote them (the branch-free lane-row's register sums, the edge samplers, backgrounds, the LUT and generic-model bodies);
    1 003 configurations and 145 three-frame launches went through offline: profiles/r05_interpreter_checksum_sweep.txt."""
    from test_emu_kernel import written_checksum
    fmt, w, h, kw = random_case(seed)
    fr = S.SyntheticFrame(fmt, w, h, **kw)
    if not _emu.fused_eligible(fr):
        pytest.skip("a frame the per-plane kernel serves")
    outs, sums = _emu.run_frames([fr], checksums=True, grid=8 if seed % 3 else 24)
    for i, (a, b) in enumerate(zip(O.run_frame(fr), outs[0])):
        assert np.array_equal(a, b), "plane %d (seed %d %s %dx%d %s)" % (i, seed, fmt, w, h, kw)
    assert sums[0] == written_checksum(fr, outs[0]), "seed %d %s %dx%d %s" % (seed, fmt, w, h, kw)
