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
