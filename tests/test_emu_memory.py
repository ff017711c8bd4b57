"""Memory safety of the kernels' address arithmetic, on the CPU tier: the host-interpreted kernels (tests/_emu.py) run with every plane placed flush
against an inaccessible page, at its end and at its start (tests/emu/guard_run.py, one process per case).  The tap fetches of the LUT samplers are
aligned dword / 16-byte reads with an optional extra dword; nothing may touch a byte outside the plane the caller declared — on the device such a read
only faults when the plane ends an allocation, which is how an overrun hides (the audit instantiation range-checks on the GPU; this is its CPU twin and
covers the per-plane kernel and every sampler too)."""
import os
import subprocess
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
RUN = os.path.join(HERE, "emu", "guard_run.py")

CASES = [("YUV422P16LE", 322, 186, 2, 1.0, 256), ("YUV422P16LE", 322, 186, 8, 1.0, 256), ("YUV422P16LE", 322, 186, 4, 1.6, 2), ("NV12", 322, 186, 2, 1.0, 256),
         ("NV12", 321, 185, 8, 2.0, 1), ("YUV420P", 322, 186, 8, 1.0, 1), ("RGBA", 201, 121, 8, 1.3, 4), ("RGB24", 201, 121, 4, 1.1, 1),
         ("RGBAF32", 201, 121, 4, 1.0, 16), ("P010LE", 322, 186, 8, 1.0, 2), ("YUVA444P10LE", 130, 70, 8, 0.8, 2)]


@pytest.mark.parametrize("place", ["end", "start"])
@pytest.mark.parametrize("which", ["fused", "plane"])
@pytest.mark.parametrize("fmt,w,h,interp,fov,align", CASES)
def test_no_byte_outside_the_planes(fmt, w, h, interp, fov, align, which, place):
    r = subprocess.run([sys.executable, RUN, fmt, str(w), str(h), str(interp), str(fov), which, place, str(align)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, "%s kernel touched memory outside a plane (%s guard): rc %d\n%s" % (which, place, r.returncode, r.stderr[-1500:])
    if r.stdout.strip().splitlines()[-1] == "SKIP":
        pytest.skip("not a frame the fused kernel serves")
    assert r.stdout.strip().splitlines()[-1] == "OK", r.stdout[-500:]
