#!/usr/bin/env python3
"""diagnosis (round 6): the GoPro certified first pass on the GPU — audit counters and the map of differing pixels for a few builds"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from gyroflow_amd import abi, synthetic as S, warp
import _oracle as O
import test_gpu_pass1_radial as T

def one(fmt, interp, w=640, h=360, fov=1.0, rl=2.5, variant=3):
    fr = S.SyntheticFrame(fmt, w, h, seed=0x6A, lens=T.gopro_lens(w, h, r_limit=rl), fov=fov, readout_ms=14.0, interpolation=interp)
    ref = O.run_frame(fr)
    backend, a, outs = T.audit(fr, variant)
    d = [x != y for x, y in zip(ref, outs)]
    y0 = d[0].reshape(fr.planes[0]["out_size"][1], -1)
    rows = np.nonzero(y0.any(axis=1))[0]
    print("%s interp %d fov %.1f rl %.1f variant %d [%s]: %s | differing bytes %s | luma rows with differences: %d (first %s, last %s)" % (
        fmt, interp, fov, rl, variant, backend, {k: (round(v, 6) if isinstance(v, float) else v) for k, v in a.items()}, [int(x.sum()) for x in d], len(rows), rows[:3], rows[-3:]), flush=True)
    got = warp.run_frame(fr, jit=2)
    print("    plain specialised build [%s]: differing bytes %s" % (warp.last_backend(), [int(np.count_nonzero(x != y)) for x, y in zip(ref, got)]), flush=True)

print("GFW_JIT_WAVES =", os.environ.get("GFW_JIT_WAVES"), " GFW_JIT_DEFS =", os.environ.get("GFW_JIT_DEFS"))
one("YUV422P16LE", 2)
one("YUV422P16LE", 8)
one("YUV422P16LE", 4)
one("NV12", 2)
one("YUV422P16LE", 2, rl=0.0)
one("YUV422P16LE", 2, fov=1.5, rl=0.0)
one("YUV422P16LE", 2, variant=4)
