#!/usr/bin/env python3
"""diagnosis (round 6): a clip launch of a specialised build whose argument block the compiler copied to scratch (generic_polynomial, certified first pass) —
which frame's picture ends up in which destination"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from gyroflow_amd import abi, synthetic as S, warp
import _oracle as O
import test_gpu_jit as J
import test_gpu_pass1_radial as T

for model in ("generic_polynomial", "sony"):
    for n in (1, 4):
        w, h = 640, 360
        frames = [S.SyntheticFrame("YUV422P16LE", w, h, seed=0x9F10 + j, timestamp_ms=1000.0 + 33.3 * j, lens=T.closed_form_lens(model, w, h, r_limit=0.0), readout_ms=16.0, pixels=False) for j in range(n)]
        backend, status, prof, outs, srcs = J.device_clip(frames, 2, True)
        refs = [O.run_frame(J._View(fr, srcs[j])) for j, fr in enumerate(frames)]
        table = [[int(np.count_nonzero(refs[i][0] != outs[j][0])) for i in range(n)] for j in range(n)]
        untouched = [int(np.count_nonzero(outs[j][0] != outs[j][0][0])) for j in range(n)]
        print(model, n, backend, status[:2], prof, "differing luma bytes [dst j][oracle i]:", table, "non-constant bytes per dst:", untouched, flush=True)
