"""How the CPU baseline (oracle/gfw_oracle.c, the C restatement of undistort_image_cpu) scales with threads on this host: one frame of YUV422P16LE through the
whole-frame entry point (one parallel region over all planes, static chunks) and through the per-plane calls, 1 thread .. all; prints what the container may use
(cpu count, affinity, cgroup quota).  usage: tools/cpu_baseline_scan.py [W H [threads,comma,separated]]   (test infrastructure: times the oracle, not the product)"""
import os, sys, time, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from gyroflow_amd import abi, synthetic as S
import _oracle as O
W, H = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (1920, 1080)
fr = S.SyntheticFrame("YUV422P16LE", W, H, seed=0x9F10)
L = O.lib()
L.gfw_oracle_undistort_frame.argtypes = [C.c_int, C.POINTER(abi.Buffers), C.POINTER(abi.KernelParams), C.POINTER(C.c_int), C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int]
outs = [pl["dst"].copy() for pl in fr.planes]
n = len(fr.planes)
bufs = (abi.Buffers * n)(*[O.host_buffers(pl["src"], pl["size"], o, pl["out_size"]) for pl, o in zip(fr.planes, outs)])
prm = (abi.KernelParams * n)(*[pl["params"] for pl in fr.planes])
typ = (C.c_int * n)(*[abi.PIXEL_TYPES[pl["pixel_type"]][0] for pl in fr.planes])
m = np.ascontiguousarray(fr.matrices, dtype=np.float32)
ref = O.run_frame(fr)
px = W * H
def t_frame(nt, chunk):
    best = 1e9
    for _ in range(3):
        a = time.perf_counter(); rc = L.gfw_oracle_undistort_frame(n, bufs, prm, typ, fr.model, fr.digital, m.ctypes.data, nt, chunk); best = min(best, time.perf_counter() - a)
        assert rc == 1
    return best
def t_planes(nt):
    best = 1e9
    for _ in range(3):
        a = time.perf_counter(); O.run_frame(fr, nthreads=nt); best = min(best, time.perf_counter() - a)
    return best
print("cpu_count", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)), "omp max", L.gfw_oracle_num_threads())
try: print("cgroup cpu.max", open("/sys/fs/cgroup/cpu.max").read().strip())
except Exception as e: print("cgroup", e)
t = t_frame(1, 4); print("frame entry, 1 thread: %.1f ms = %.2f Mpix/s" % (t * 1e3, px / t / 1e6))
assert all(np.array_equal(a, b) for a, b in zip(ref, outs)), "frame entry differs from per-plane calls"
os.environ["GFW_ORACLE_GENERIC"] = "1"; t = t_frame(1, 4); print("generic form, 1 thread: %.1f ms = %.2f Mpix/s" % (t * 1e3, px / t / 1e6)); del os.environ["GFW_ORACLE_GENERIC"]
for nt in [int(x) for x in (sys.argv[3] if len(sys.argv) > 3 else "2,4,8").split(",")]:
    tf = t_frame(nt, 4); tp = t_planes(nt)
    print("threads %3d: frame entry %.1f ms = %.1f Mpix/s (%.2f per thread) | per-plane calls %.1f ms = %.1f Mpix/s" % (nt, tf * 1e3, px / tf / 1e6, px / tf / 1e6 / nt, tp * 1e3, px / tp / 1e6))
