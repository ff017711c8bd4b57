import sys
import os; ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import _emu
from test_emu_pass1_audit import random_clip
worst = 0.0; tot = [0, 0, 0]; used = 0; bad = 0
for seed in range(int(sys.argv[1]), int(sys.argv[2])):
    fr = random_clip(seed)
    p0 = fr.planes[0]["params"]
    if _emu.p1_table(p0, fr.matrices, p0.matrix_count) is None: continue
    outs, a = _emu.run_frames([fr], audit=True)
    used += 1; tot[0] += a["certified"]; tot[1] += a["queued"]; tot[2] += p0.output_width * p0.output_height
    r = a["gap_px"] / a["eps_px"]; worst = max(worst, r)
    if a["wrong"] or a["queue_overflow"] or a["out_of_range"] or r >= 0.5: bad += 1; print("BAD seed", seed, a, flush=True)
    if seed % 50 == 0: print("... seed", seed, "used", used, "worst gap/E %.3f" % worst, "bad", bad, flush=True)
print("done used", used, "certificates", tot[0], "queued", tot[1], "pixels", tot[2], "worst gap/E %.3f" % worst, "bad", bad)
