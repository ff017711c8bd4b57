import sys, time
import os; ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from gyroflow_amd import synthetic as S
import _emu, _oracle as O
from test_gpu_fuzz import random_case
from test_emu_kernel import written_checksum
lo, hi = int(sys.argv[1]), int(sys.argv[2])
served = bad = 0
t0 = time.time()
for seed in range(lo, hi):
    fmt, w, h, kw = random_case(seed)
    kw = dict(kw); kw.pop("seed", None); kw.pop("timestamp_ms", None)
    try:
        frames = [S.SyntheticFrame(fmt, w, h, seed=seed * 7 + j, timestamp_ms=1000.0 + 33.3 * j, **kw) for j in range(3)]
    except TypeError as e:
        print("seed", seed, "kw", repr(e)[:200]); continue
    if not _emu.fused_eligible(frames[0]):
        continue
    try:
        outs, sums = _emu.run_frames(frames, checksums=True, grid=(8, 16, 64)[seed % 3])
    except Exception as e:                     # noqa: BLE001
        print("seed", seed, fmt, w, h, "EXC", repr(e)[:300]); bad += 1; continue
    served += 1
    for j, fr in enumerate(frames):
        ok_px = all(np.array_equal(a, b) for a, b in zip(O.run_frame(fr), outs[j]))
        ok_ck = sums[j] == written_checksum(fr, outs[j])
        if not (ok_px and ok_ck):
            bad += 1
            print("seed", seed, "frame", j, fmt, w, h, kw, "pixels", ok_px, "checksum", ok_ck)
print("seeds %d..%d: %d three-frame launches through the checksum build, %d bad, %.0f s" % (lo, hi, served, bad, time.time() - t0))
