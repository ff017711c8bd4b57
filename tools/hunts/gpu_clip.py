# GPU hunt (round 6): the fuzz generator's configurations as CLIP launches of 2-5 frames of the clip's specialised build on the device (gfw_undistort_clip, device-resident
# planes and matrix tables) against the oracle, frame by frame — what the interpreter's clip hunt (clip.py) cannot see: the compiler.  usage: gpu_clip.py A B
import sys, copy, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from gyroflow_amd import synthetic as S, warp
import _oracle as O
from test_gpu_fuzz import random_case
import test_gpu_jit as J
a0, a1 = int(sys.argv[1]), int(sys.argv[2])
bad = used = fused = 0
backends = {}
t0 = time.time()
for seed in range(a0, a1):
    try:
        fmt, w, h, kw = random_case(seed)
        if kw["interpolation"] > 8: continue                      # EWA: the per-plane kernel frame by frame (tests/test_gpu_fuzz.py covers it); the oracle takes seconds per frame
        n = 2 + seed % 4
        frames = []
        for j in range(n):
            k2 = copy.deepcopy(kw); k2["seed"] = kw["seed"] + 17 * j; k2["timestamp_ms"] = 1000.0 + 33.3 * j
            frames.append(S.SyntheticFrame(fmt, w, h, pixels=False, **k2))
        used += 1
        backend, status, prof, outs, srcs = J.device_clip(frames, 2, True)
        backends[backend] = backends.get(backend, 0) + 1
        fused += backend.startswith("yuv_fused")
        for j, fr in enumerate(frames):
            d = [int(np.count_nonzero(np.asarray(a) != np.asarray(b))) for a, b in zip(O.run_frame(J._View(fr, srcs[j])), outs[j])]
            if any(d): bad += 1; print("MISMATCH seed", seed, fmt, w, h, backend, "frame", j, "of", n, d, kw, flush=True)
    except Exception as e:
        bad += 1; print("ERROR seed", seed, repr(e)[:300], flush=True)
    if seed % 50 == 0: print("... seed", seed, "used", used, "fused", fused, "bad", bad, "%.0f s" % (time.time() - t0), flush=True)
print("done", a0, a1, "used", used, "fused", fused, "bad", bad, backends, "%.0f s" % (time.time() - t0))
