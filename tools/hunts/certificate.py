# first-pass certificate hunt: random fisheye clips (lens coefficients, focal length, field of view, readout, shutter direction, sizes) through the
# interpreted fused kernel vs the oracle: a wrong certificate (approximate row != exact row) shows up as differing pixels
import sys
import os; ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from gyroflow_amd import synthetic as S
import _emu, _oracle as O
a0, a1 = int(sys.argv[1]), int(sys.argv[2])
bad = 0; used = 0
for seed in range(a0, a1):
    rng = np.random.default_rng(50000 + seed)
    w = int(rng.integers(80, 700)) * 2; h = int(rng.integers(60, 400)) * 2
    lens = S.gopro_style_lens(w, h)
    lens["f"] = (float(rng.uniform(0.3, 1.2)) * w,) * 2
    lens["c"] = (w / 2.0 + float(rng.uniform(-0.05, 0.05)) * w, h / 2.0 + float(rng.uniform(-0.05, 0.05)) * h)
    lens["k"] = [float(rng.uniform(-0.08, 0.12)), float(rng.uniform(-0.05, 0.05)), float(rng.uniform(-0.03, 0.03)), float(rng.uniform(-0.01, 0.01))] + [0.0] * 8
    kw = dict(seed=int(rng.integers(1, 1 << 20)), lens=lens, fov=float(rng.uniform(0.5, 3.0)), readout_ms=float(rng.uniform(-30.0, 30.0)),
              horizontal_rs=bool(rng.random() < 0.3), interpolation=2)
    try:
        fr = S.SyntheticFrame("NV12", w, h, **kw)
        p0 = fr.planes[0]["params"]
        if _emu.p1_table(p0, fr.matrices, p0.matrix_count) is None: continue
        used += 1
        ref = O.run_frame(fr); got = _emu.run_frame(fr)
        n = [int(np.count_nonzero(a != b)) for a, b in zip(ref, got)]
        if any(n): bad += 1; print("MISMATCH seed", seed, w, h, n, kw, flush=True)
    except Exception as e:
        bad += 1; print("ERROR seed", seed, repr(e)[:300], flush=True)
    if seed % 50 == 0: print("... seed", seed, "used", used, "bad", bad, flush=True)
print("done", a0, a1, "used", used, "bad", bad)
