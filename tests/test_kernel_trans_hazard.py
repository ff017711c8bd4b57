"""gfx950 needs a wait state between a transcendental instruction (v_rcp / v_rsq / v_sqrt / v_exp / v_log / v_sin / v_cos) and a VALU instruction that reads its
result.  The compiler inserts it for its own instructions; it cannot see inside an `asm` statement.  Round 6 met the pattern for real: the r-form first pass took
`v_sqrt_f32` and clamped the result with min_limit's inline `v_min_f32` — on the MI355X a third of the certificates of a GoPro clip came out wrong (gaps of 39 px, varying
with the optimisation level) while the interpreter, which has no pipeline, was clean (profiles/r06_gopro_first_pass.txt).  tools/scan_trans_hazard.py reads the device
code for the pattern; here it reads every code object the library ships and the specialised builds of a gopro clip (compiled on the host, no device needed)."""
import ctypes as C
import os
import subprocess
import sys

import pytest

from gyroflow_amd import abi, synthetic as S

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SCAN = os.path.join(ROOT, "tools", "scan_trans_hazard.py")


def test_shipped_device_code_has_no_transcendental_read_by_the_next_instruction():
    r = subprocess.run([sys.executable, SCAN], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "0 place(s)" in r.stdout, r.stdout[-3000:] + r.stderr[-2000:]


@pytest.mark.parametrize("interp", [2, 8])
def test_the_r_form_first_pass_of_a_gopro_clip_is_hazard_free(tmp_path, interp):
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import build_jit_cache as B
    lib = abi.load_library()
    lib.gfw_debug_jit_compile.argtypes = [C.c_char_p, C.c_char_p, C.c_char_p, C.c_char_p, C.c_char_p, C.c_size_t]
    lib.gfw_debug_jit_compile.restype = C.c_long
    lens = dict(S.gopro_style_lens(1920, 1080))
    lens["model"], lens["k"], lens["r_limit"] = "gopro", [0.0, 1.0, 0.01, -0.12, 0.02, 0.01, -0.004] + [0.0] * 5, 2.5
    fr = S.SyntheticFrame("YUV422P16LE", 1920, 1080, seed=3, lens=lens, pixels=False, interpolation=interp)
    defs, header, _ = B.key_of(lib, fr)
    assert b"GFW_JIT_FAST1=1" in defs and b"#define GFW_P1_RFORM (1)" in header           # the certified pass, table over r
    out = str(tmp_path / "gopro.co")
    log = C.create_string_buffer(1 << 16)
    n = lib.gfw_debug_jit_compile(b"gfx950", defs, header, out.encode(), log, len(log))
    if n == -2:
        pytest.skip("libhiprtc.so not found")
    assert n > 0, log.value.decode()[-2000:]
    r = subprocess.run([sys.executable, SCAN, out], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "0 place(s)" in r.stdout, r.stdout[-2000:]
    dis = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-objdump", "-d", out], capture_output=True, text=True).stdout
    assert "v_sqrt_f32" in dis                                                              # (the r form is in there: the scan had something to look at)
