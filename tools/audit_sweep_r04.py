#!/usr/bin/env python3
"""Round 4: a wider audit of the derived first-pass certificate on the device — N seeded random clips (tests/test_gpu_pass1_sweep.py's generator: lens, focal
length, field of view 0.5-3, rotation up to 15 degrees, readout +-30 ms at up to 250 deg/s, both shutter directions) with, in turn, an r_limit, coordinates moved
away from the origin (translation2d up to 3e4 px, matrices compensated: cancellation in the linear forms) and both; every certificate of every frame is re-derived by
the audit instantiation.  Prints one summary; exit status 1 on any wrong certificate or a gap beyond E.   usage: tools/audit_sweep_r04.py [N]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_gpu_pass1_sweep as T  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 600
rng = np.random.default_rng(0xCE47)
served = declined = nothing = 0
pixels = certified = 0
worst = (0.0, None)
bad = []
for i in range(N):
    w, h = T.SIZES[int(rng.integers(0, len(T.SIZES)))]
    fr = T.random_clip(rng, w, h)
    kind = i % 4
    shift = 0.0
    if kind in (1, 3):
        for pl in fr.planes:
            pl["params"].r_limit = float(rng.uniform(0.4, 3.0))
    if kind in (2, 3):
        shift = float(rng.choice([100.0, 1000.0, 10000.0, 30000.0]))
        sx, sy = shift * float(rng.choice([-1.0, 1.0])), shift * float(rng.choice([-1.0, 1.0]))
        for p, pl in enumerate(fr.planes):
            pl["params"].translation2d[0], pl["params"].translation2d[1] = sx, sy
        m = fr.matrices
        for col in (0, 3, 6):
            m[:, col + 2] -= np.float32(sx) * m[:, col] + np.float32(sy) * m[:, col + 1]
    backend, a = T.audit_device(fr)
    if backend != "yuv_fused_p1":
        declined += 1
        continue
    served += 1
    if a["certified1_wrong"] or a["out_of_range"] or a["queue_overflow"] or a["certified1"] + a["queued1"] != w * h:
        bad.append((i, kind, shift, w, h, a))
    if a["certified1"] == 0:
        nothing += 1
        continue
    pixels += w * h
    certified += a["certified1"]
    ratio = a["pass1_gap_px"] / a["pass1_eps_px"] if a["pass1_eps_px"] > 0 else float("inf")
    if not ratio < 1.0:
        bad.append((i, kind, shift, w, h, a))
    if ratio > worst[0]:
        worst = (ratio, (i, kind, shift, w, h, a["pass1_gap_px"], a["pass1_eps_px"]))
print("clips %d: served by the certified kernel %d (declined by the host %d), of those %d frames certify nothing (E >= 0.2 px / r-limit margin / W range)" % (N, served, declined, nothing))
print("pixels audited %d, certified %.2f %%, wrong certificates %d, worst gap / E = %.3f at %s" % (pixels, 100.0 * certified / max(pixels, 1), sum(b[5]["certified1_wrong"] for b in bad), worst[0], worst[1]))
for b in bad[:10]:
    print("BAD", b)
sys.exit(1 if bad else 0)
