"""The C++ host-side mirror (include/gfwarp.hpp: gyroflow::Stabilization, Buffers, FrameTransform, GyroflowCoreError)
driven by a C++ program the way the reference's render loop drives its Stabilization (tests/cpp/test_operator.cpp).

CPU: `process_pixels` validation order and error variants (mod.rs:612-640), `get_frame_transform_at` fill-in, and the loud
failure without a device.  GPU: Luma16 fisheye + rolling-shutter warps through the HIP backend, bit-exact vs the oracle."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIBDIR = os.path.join(ROOT, "gyroflow_amd")


@pytest.fixture(scope="module")
def exe(tmp_path_factory):
    if not os.path.exists(os.path.join(LIBDIR, "libgfwarp.so")):
        pytest.skip("libgfwarp.so not built")
    out = str(tmp_path_factory.mktemp("cpp") / "test_operator")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-Wall", "-Wextra", "-Werror", "-I" + os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "cpp", "test_operator.cpp"), "-o", out,
                           "-L" + LIBDIR, "-lgfwarp", "-ldl", "-Wl,-rpath," + LIBDIR])
    return out


def test_cpp_operator_validation_and_fill_in(exe):
    out = subprocess.run([exe, "validate"], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "validate ok" in out.stdout


@pytest.mark.gpu
def test_cpp_operator_warp_bit_exact(exe):
    import _oracle as O
    O.lib()                                                   # builds oracle/libgfw_oracle.so when stale
    out = subprocess.run([exe, "warp", O.ORACLE_SO], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "warp ok" in out.stdout and out.stdout.count("0 differing bytes") == 3
