"""TEST INFRASTRUCTURE (run by tests/test_emu_memory.py, one process per case): every source and destination plane of a frame, its matrix table and the first pass's table are placed flush against an
inaccessible page — `end`: the plane's last byte is the last accessible one, `start`: its first byte the first — and the frame goes through the
host-interpreted kernels (tests/_emu.py).  A single byte read or written outside a plane kills this process with SIGSEGV; otherwise it prints OK (and
the result still equals the oracle's).
usage: guard_run.py <format> <width> <height> <interpolation> <fov> fused|plane end|start [stride_align]   |   guard_run.py --seed N fused|plane end|start"""
import ctypes as C
import mmap
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from gyroflow_amd import synthetic as S
import _emu, _oracle as O
guarded = _emu.guarded
if sys.argv[1] == "--seed":           # guard_run.py --seed N fused|plane end|start: configuration N of tests/test_gpu_fuzz.py's generator
    from test_gpu_fuzz import random_case
    fmt, w, h, kw = random_case(int(sys.argv[2]))
    which, at_end = sys.argv[3], sys.argv[4] == "end"
    fr = S.SyntheticFrame(fmt, w, h, **kw)
else:
    fmt, w, h, interp, fov, which, at_end = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), float(sys.argv[5]), sys.argv[6], sys.argv[7] == "end"
    fr = S.SyntheticFrame(fmt, w, h, seed=0x33, fov=fov, interpolation=interp, stride_align=int(sys.argv[8]) if len(sys.argv) > 8 else 256)
if which == "fused" and not _emu.fused_eligible(fr):
    print("SKIP")
    sys.exit(0)
ref = O.run_frame(fr)
for pl in fr.planes:
    pl["src"] = guarded(pl["src"], at_end)
    pl["dst"] = guarded(pl["dst"], at_end)
_emu.GUARD = "end" if at_end else "start"
got = _emu.run_frame(fr) if which == "fused" else _emu.run_frame_per_plane(fr)
ok = all(np.array_equal(a, b) for a, b in zip(ref, got))
print("OK" if ok else "MISMATCH")
