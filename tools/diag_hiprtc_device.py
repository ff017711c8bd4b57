#!/usr/bin/env python3
"""Diagnosis (round 4): does hiprtc's output depend on whether the process has initialised a HIP device?  The build step's kernel cache is compiled device-less;
the run-time path compiles with a device current — for C2 bicubic the two binaries differ (25 216 vs 25 408 bytes, 78 vs 64 us).  Compiles the same key twice in one
process, before and after gfw_set_device(0), with comgr's verbose log redirected per phase.  usage: tools/diag_hiprtc_device.py outdir"""
import ctypes as C
import hashlib
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
out = sys.argv[1]
os.makedirs(out, exist_ok=True)
os.environ["AMD_COMGR_EMIT_VERBOSE_LOGS"] = "1"
os.environ["AMD_COMGR_REDIRECT_LOGS"] = os.path.join(out, "comgr_before.log")
import build_jit_cache as B  # noqa: E402
from gyroflow_amd import abi  # noqa: E402
lib = abi.load_library()
defs, header, name = B.key_of(lib, B.bench_frame(interp=4))
lib.gfw_debug_jit_compile.argtypes = [C.c_char_p] * 4 + [C.c_char_p, C.c_size_t]
lib.gfw_debug_jit_compile.restype = C.c_long


def comp(tag):
    log = C.create_string_buffer(1 << 16)
    p = os.path.join(out, tag + ".co")
    n = lib.gfw_debug_jit_compile(B.ARCH, defs, header, p.encode(), log, len(log))
    h = hashlib.sha1(open(p, "rb").read()).hexdigest()[:12] if n > 0 else "-"
    print(tag, n, h, flush=True)


comp("before_device")
os.environ["AMD_COMGR_REDIRECT_LOGS"] = os.path.join(out, "comgr_after.log")
print("gfw_set_device ->", lib.gfw_set_device(0), flush=True)
import torch  # noqa: E402
torch.zeros(4, device="cuda:0").sum().item()
comp("after_device")
shipped = os.path.join(ROOT, "gyroflow_amd", "jit_cache", name)
print("shipped", hashlib.sha1(open(shipped, "rb").read()).hexdigest()[:12] if os.path.exists(shipped) else "absent")
