import sys
import os; ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from gyroflow_amd import synthetic as S
import _emu, _oracle as O
bad = 0
for fmt, w, h, n in (("YUV422P16LE", 640, 360, 1), ("YUV422P16LE", 384, 208, 5), ("NV12", 642, 362, 3), ("YUV420P", 1280, 720, 2), ("RGBA", 200, 120, 16), ("YUV422P16LE", 130, 70, 7)):
    frames = [S.SyntheticFrame(fmt, w, h, seed=0x9F10 + j, timestamp_ms=1000.0 + 33.3 * j) for j in range(n)]
    refs = [O.run_frame(f) for f in frames]
    for grid in (8, 16, 24, 64, 256, 2048):
        outs = _emu.run_frames(frames, grid=grid)
        ok = all(np.array_equal(a, b) for r, o in zip(refs, outs) for a, b in zip(r, o))
        bad += not ok
        print(fmt, w, h, "frames", n, "grid", grid, "OK" if ok else "MISMATCH", flush=True)
print("bad", bad)
