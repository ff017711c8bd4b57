# GPU hunt (round 6): the fuzz generator's configurations, EWA included, as single frames through gfw_undistort_frame on the device (HOST buffers, default options) against
# the oracle — what test_gpu_fuzz.py does for its 80 seeds, at length.  Reports which backend served how many, and how many frames put two planes into one launch
# (EWA on planar chroma: gfw_plane_kernel<.., DUAL>).  usage: gpu_frame.py A B
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from gyroflow_amd import abi, synthetic as S, warp
import _oracle as O
from test_gpu_fuzz import random_case
a0, a1 = int(sys.argv[1]), int(sys.argv[2])
lib = abi.load_library()
bad = used = paired = ewa = 0
backends = {}
t0 = time.time()
for seed in range(a0, a1):
    try:
        fmt, w, h, kw = random_case(seed)
        fr = S.SyntheticFrame(fmt, w, h, **kw)
        ref = O.run_frame(fr)
        outs = [pl["dst"].copy() for pl in fr.planes]
        bufs = [warp.host_buffers(pl["src"], pl["size"], o, pl["out_size"]) for pl, o in zip(fr.planes, outs)]
        params, types = [pl["params"] for pl in fr.planes], [pl["pixel_type"] for pl in fr.planes]
        be = warp.Backend(params[0], types[0], fr.model, fr.digital, bufs[0])
        try:
            be.undistort_frame(bufs, params, types, fr.matrices)
            paired += int(lib.gfw_debug_paired_launches(be.ctx))
            backend = warp.last_backend()
        finally:
            be.close()
        used += 1
        ewa += kw["interpolation"] > 8
        backends[backend] = backends.get(backend, 0) + 1
        d = [int(np.count_nonzero(np.asarray(a) != np.asarray(b))) for a, b in zip(ref, outs)]
        if any(d): bad += 1; print("MISMATCH seed", seed, fmt, w, h, backend, d, kw, flush=True)
    except Exception as e:
        print("ERROR seed", seed, repr(e)[:300], flush=True)
    if (seed - a0) % 100 == 99: print("...", seed, "bad", bad, flush=True)
print("done", a0, a1, "used", used, "ewa", ewa, "paired launches", paired, "bad", bad, backends, "%.0f s" % (time.time() - t0), flush=True)
