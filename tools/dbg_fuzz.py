import sys
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np
from gyroflow_amd import synthetic as S, warp
import _oracle as O
from test_gpu_fuzz import random_case
seed = int(sys.argv[1])
fmt, w, h, kw = random_case(seed)
fr = S.SyntheticFrame(fmt, w, h, **kw)
ref = O.run_frame(fr)
got = warp.run_frame(fr); bk = warp.last_backend()
gen = warp.run_frame(fr, fused=False)
for i, pl in enumerate(fr.planes):
    ow, oh, ostride = pl["out_size"]
    bpp = pl["params"].bytes_per_pixel
    a = ref[i].reshape(-1, ostride); b = got[i].reshape(-1, ostride); g = gen[i].reshape(-1, ostride)
    bad = np.argwhere(a != b)
    print("plane", i, bk, "fused-vs-oracle diffs", len(bad), "generic-vs-oracle diffs", int((a != g).sum()))
    seen = set()
    for y, xb in bad[:40]:
        x = xb // bpp
        if (x, y) in seen: continue
        seen.add((x, y))
        ok, u, v = O.undistort_coord(pl["params"], fr.model, fr.digital, fr.matrices, float(x), float(y))
        print("  px", x, y, "oracle coord", ok, u, v, "ref", a[y, x*bpp:(x+1)*bpp], "got", b[y, x*bpp:(x+1)*bpp])
