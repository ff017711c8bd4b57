#!/usr/bin/env python3
"""diagnosis (round 6): the certified first pass of the closed-form radial models on the GPU — audit counters and where the frames differ from the oracle"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from gyroflow_amd import abi, synthetic as S, warp
import _oracle as O
import test_gpu_pass1_radial as T

def one(model, fmt, w, h, fov, rl, seed, variant=3, k=None, interp=2):
    fr = S.SyntheticFrame(fmt, w, h, seed=seed, lens=T.closed_form_lens(model, w, h, r_limit=rl, k=k), fov=fov, readout_ms=14.0, interpolation=interp)
    ref = O.run_frame(fr)
    for label, run in (("audit %d" % variant, lambda: T.audit(fr, variant)), ("plain", lambda: (None, None, warp.run_frame(fr, jit=2)))):
        backend, a, outs = run()
        backend = backend or warp.last_backend()
        d = [np.flatnonzero(np.asarray(x) != np.asarray(y)) for x, y in zip(ref, outs)]
        bps = 2 if "16" in fmt or fmt == "P010" else 1
        pw = fr.planes[0]["out_size"][2] // 1
        where = [(int(b) // pw, (int(b) % pw) // bps) for b in d[0][:6]]
        print("%s %s %dx%d fov %.1f rl %.1f %s [%s]: %s | differing bytes %s | first luma (row, col): %s" % (
            model, fmt, w, h, fov, rl, label, backend, {k_: (round(v, 6) if isinstance(v, float) else v) for k_, v in (a or {}).items()}, [int(x.size) for x in d], where), flush=True)

import bench
K = bench.LENS_MODEL_K
one("ptlens", "YUV422P16LE", 960, 540, 1.0, 2.5, 0x70 + 10)
one("ptlens", "YUV422P16LE", 960, 540, 1.0, 2.5, 0x70 + 10, variant=4)
one("generic_polynomial", "YUV422P16LE", 3840, 2160, 1.0, 0.0, 0x9F10, k=K["generic_polynomial"])
one("generic_polynomial", "YUV422P16LE", 1920, 1080, 1.0, 0.0, 0x9F10, k=K["generic_polynomial"])
one("generic_polynomial", "YUV422P16LE", 960, 540, 1.0, 2.5, 0x70 + 10)
one("sony", "YUV422P16LE", 3840, 2160, 1.0, 0.0, 0x9F10)
one("poly5", "YUV422P16LE", 960, 540, 1.5, 0.0, 0x70 + 15)
