# Interpreter hunt (round 6): the fuzz generator's EWA configurations on planar formats — U and V through ONE host-interpreted launch of gfw_plane_kernel<.., DUAL>
# (tests/_emu.py run_plane_pair) against the oracle plane by plane.  usage: ewa_pair.py A B      (round 6: seeds 50000..50599, 88 configurations, 0 mismatches)
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from gyroflow_amd import synthetic as S
import _oracle as O, _emu
from test_gpu_fuzz import random_case
a0,a1=int(sys.argv[1]),int(sys.argv[2])
used=bad=0
for seed in range(a0,a1):
    fmt,w,h,kw=random_case(seed)
    if kw["interpolation"]<=8: continue
    fr=S.SyntheticFrame(fmt,w,h,**kw)
    if len(fr.planes)<3 or fr.planes[1]["pixel_type"]!=fr.planes[2]["pixel_type"]: continue
    pa,pb=fr.planes[1]["params"],fr.planes[2]["params"]
    used+=1
    got=_emu.run_plane_pair(fr,1)
    for k,i in enumerate((1,2)):
        pl=fr.planes[i]; ref=pl["dst"].copy()
        assert O.undistort_image(pl["src"],pl["size"],ref,pl["out_size"],pl["params"],pl["pixel_type"],fr.model,fr.digital,fr.matrices)==1
        if not np.array_equal(ref,got[k]): bad+=1; print("MISMATCH",seed,fmt,w,h,i,int(np.count_nonzero(ref!=got[k])),kw,flush=True)
print("done",a0,a1,"used",used,"bad",bad,flush=True)
