"""one-off: the random sweep's configurations through the checksum build of the interpreted fused kernel (sums vs the written bytes, pixels vs the oracle)"""
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
    fr = S.SyntheticFrame(fmt, w, h, **kw)
    if not _emu.fused_eligible(fr):
        continue
    try:
        outs, sums = _emu.run_frames([fr], checksums=True, grid=8 if seed % 3 else 24)
    except Exception as e:                     # noqa: BLE001
        print("seed", seed, fmt, w, h, "EXC", repr(e)[:300]); bad += 1; continue
    served += 1
    ref = O.run_frame(fr)
    ok_px = all(np.array_equal(a, b) for a, b in zip(ref, outs[0]))
    ok_ck = sums[0] == written_checksum(fr, outs[0])
    if not (ok_px and ok_ck):
        bad += 1
        print("seed", seed, fmt, w, h, kw, "pixels", ok_px, "checksum", ok_ck)
print("seeds %d..%d: %d through the checksum build, %d bad, %.0f s" % (lo, hi, served, bad, time.time() - t0))
