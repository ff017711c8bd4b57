import sys, traceback
import os; ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from gyroflow_amd import synthetic as S
import _emu, _oracle as O
from test_gpu_fuzz import random_case
a0, a1 = int(sys.argv[1]), int(sys.argv[2])
bad = 0
for seed in range(a0, a1):
    try:
        fmt, w, h, kw = random_case(seed)
        fr = S.SyntheticFrame(fmt, w, h, **kw)
        ref = O.run_frame(fr)
        for i, (a, b) in enumerate(zip(ref, _emu.run_frame_per_plane(fr))):
            if not np.array_equal(a, b): bad += 1; print("MISMATCH per-plane seed", seed, fmt, w, h, "plane", i, int(np.count_nonzero(a != b)), kw, flush=True)
        if _emu.fused_eligible(fr):
            for i, (a, b) in enumerate(zip(ref, _emu.run_frame(fr))):
                if not np.array_equal(a, b): bad += 1; print("MISMATCH fused seed", seed, fmt, w, h, "plane", i, int(np.count_nonzero(a != b)), kw, flush=True)
    except Exception as e:
        bad += 1; print("ERROR seed", seed, repr(e)[:300], flush=True)
    if seed % 100 == 0: print("... seed", seed, "bad", bad, flush=True)
print("done", a0, a1, "bad", bad)
