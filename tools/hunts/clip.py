# clip-launch hunt: the fuzz generator's fused-eligible configurations as launches of 2-5 frames (different seeds / timestamps per frame, same constants)
import sys, copy
import os; ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from gyroflow_amd import synthetic as S
import _emu, _oracle as O
from test_gpu_fuzz import random_case
a0, a1 = int(sys.argv[1]), int(sys.argv[2])
bad = 0; used = 0
for seed in range(a0, a1):
    try:
        fmt, w, h, kw = random_case(seed)
        fr = S.SyntheticFrame(fmt, w, h, **kw)
        if not _emu.fused_eligible(fr): continue
        n = 2 + seed % 4
        frames = []
        for j in range(n):
            k2 = copy.deepcopy(kw); k2["seed"] = kw["seed"] + 17 * j; k2["timestamp_ms"] = 1000.0 + 33.3 * j
            frames.append(S.SyntheticFrame(fmt, w, h, **k2))
        # a launch needs the same constants for every frame: matrix tables and pixels differ, the feature bits must not
        if len(set(_emu.feature_bits(f) for f in frames)) != 1: continue
        used += 1
        outs = _emu.run_frames(frames)
        for j, (f, got) in enumerate(zip(frames, outs)):
            d = [int(np.count_nonzero(a != b)) for a, b in zip(O.run_frame(f), got)]
            if any(d): bad += 1; print("MISMATCH seed", seed, fmt, w, h, "frame", j, d, kw, flush=True)
    except Exception as e:
        bad += 1; print("ERROR seed", seed, repr(e)[:300], flush=True)
    if seed % 100 == 0: print("... seed", seed, "used", used, "bad", bad, flush=True)
print("done", a0, a1, "used", used, "bad", bad)
