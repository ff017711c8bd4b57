#!/usr/bin/env python3
"""Diagnosis (round 4, C2 bicubic: one binary, 78 us loaded from the kernel cache, 64 us compiled in the process): bench.py in THIS process after a prelude —
`compile`: a hiprtc build through gfw_debug_jit_compile (libhiprtc / comgr loaded, ~300 ms of host work, result discarded); `sleep`: 400 ms of nothing.
usage: tools/bench_with_prelude.py compile|sleep|none <bench.py arguments>"""
import ctypes as C
import os
import runpy
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
mode = sys.argv[1]
if mode == "compile":
    import build_jit_cache as B
    from gyroflow_amd import abi
    lib = abi.load_library()
    defs, header, name = B.key_of(lib, B.bench_frame(interp=4))
    lib.gfw_debug_jit_compile.argtypes = [C.c_char_p] * 4 + [C.c_char_p, C.c_size_t]
    lib.gfw_debug_jit_compile.restype = C.c_long
    log = C.create_string_buffer(1 << 16)
    t0 = time.time()
    n = lib.gfw_debug_jit_compile(B.ARCH, defs, header, b"", log, len(log))
    sys.stderr.write("prelude: compiled %d bytes in %.0f ms\n" % (n, (time.time() - t0) * 1e3))
elif mode == "sleep":
    time.sleep(0.4)
sys.argv = [os.path.join(ROOT, "bench.py")] + sys.argv[2:]
runpy.run_path(sys.argv[0], run_name="__main__")
