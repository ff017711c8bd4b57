# the fused-eligible configurations of the fuzz generator through the interpreter's other modes: the ahead-of-time form of the body (GFW_BAKE = 0), every wave vote
# answered pessimistically, the hardware approximations moved by one ulp either way
import sys
import os; ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from gyroflow_amd import synthetic as S
import _emu, _oracle as O
from test_gpu_fuzz import random_case
a0, a1 = int(sys.argv[1]), int(sys.argv[2])
bad = used = 0
for seed in range(a0, a1):
    try:
        fmt, w, h, kw = random_case(seed)
        fr = S.SyntheticFrame(fmt, w, h, **kw)
        if not _emu.fused_eligible(fr): continue
        used += 1
        ref = O.run_frame(fr)
        for tag, kwargs in (("aot", dict(baked=False)), ("votes", dict(votes=1)), ("ulp+1", dict(hw_ulp=1)), ("ulp-1", dict(hw_ulp=-1))):
            got = _emu.run_frames([fr], **kwargs)[0]
            d = [int(np.count_nonzero(a != b)) for a, b in zip(ref, got)]
            if any(d): bad += 1; print("MISMATCH", tag, "seed", seed, fmt, w, h, d, kw, flush=True)
    except Exception as e:
        bad += 1; print("ERROR seed", seed, repr(e)[:300], flush=True)
    if seed % 100 == 0: print("... seed", seed, "used", used, "bad", bad, flush=True)
print("done", a0, a1, "used", used, "bad", bad)
