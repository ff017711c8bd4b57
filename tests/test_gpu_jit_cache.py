"""The shipped cache of specialised kernels (gyroflow_amd/jit_cache, tools/build_jit_cache.py): the BASELINE configurations' kernels are built with the library and
found next to it, so the headline rate does not depend on libhiprtc.so being installed where the library runs.  GFW_NO_HIPRTC=1 makes the
library behave like a box without hiprtc: the C1 frame (1080p NV12, bench.py --c1) must still run through its specialised kernel, bit-exact; a clip the cache
does not hold runs ahead of time and says why."""
import os
import subprocess
import sys
import textwrap

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SCRIPT = textwrap.dedent("""
    import os, sys
    sys.path.insert(0, %r); sys.path.insert(0, os.path.join(%r, "tests")); sys.path.insert(0, os.path.join(%r, "tools"))
    import numpy as np, torch
    from gyroflow_amd import abi, synthetic as S, warp
    import _oracle as O
    import build_jit_cache as B
    from test_gpu_fullsize import _View
    dev = torch.device("cuda", 0)
    assert abi.load_library().gfw_set_device(0) == 0
    def run(fr):
        fr = S.SyntheticFrame(fr.fmt, fr.planes[0]["params"].width, fr.planes[0]["params"].height, seed=0x9F10, timestamp_ms=1000.0, lens=dict(fr.lens), fov=fr.fov if hasattr(fr, "fov") else 1.0,
                              interpolation=fr.planes[0]["params"].interpolation, readout_ms=0.0 if CQ is not None else 16.0, constant_quat=CQ) if False else fr
        d_src, d_dst = fr.device_planes(dev), fr.device_outputs(dev)
        d_mat = torch.from_numpy(warp.pack_matrices(fr.matrices)).to(dev)
        torch.cuda.synchronize(dev)                          # (torch produced the planes on ITS stream; the context's stream is non-blocking)
        bufs = [warp.device_buffers(d_src[p].data_ptr(), d_src[p].numel(), pl["size"], d_dst[p].data_ptr(), d_dst[p].numel(), pl["out_size"]) for p, pl in enumerate(fr.planes)]
        be = warp.Backend(fr.planes[0]["params"], fr.planes[0]["pixel_type"], fr.model, fr.digital, bufs[0])
        be.set_option(abi.OPT_SYNCHRONOUS, 0); be.set_option(abi.OPT_MATRICES_ON_DEVICE, 2); be.set_option(abi.OPT_JIT, 2)
        be.undistort_frame(bufs, [pl["params"] for pl in fr.planes], [pl["pixel_type"] for pl in fr.planes], d_mat.data_ptr(), matrix_count=fr.matrices.shape[0])
        be.synchronize()
        name, st = warp.last_backend(), be.jit_status()
        ref = O.run_frame(_View(fr, [t.cpu().numpy() for t in d_src]))
        same = all(np.array_equal(a, b.cpu().numpy()) for a, b in zip(ref, d_dst))
        be.close()
        return name, st, same
    CQ = None
    kw = dict(B.CONFIGS)["C1 1080p NV12 constant quaternion"]
    fr = B.bench_frame(**kw)
    fr = S.SyntheticFrame("NV12", 1920, 1080, seed=0x9F10, timestamp_ms=1000.0, lens=dict(S.gopro_style_lens(1920, 1080)), fov=1.0, interpolation=2, readout_ms=0.0,
                          constant_quat=S.quat_from_euler_deg(5.0, 2.0, 3.0))                       # the same frame with its pixels
    name, st, same = run(fr)
    print("C1", name, st[0], st[2][:60].replace("\\n", " "), same)
    other = S.SyntheticFrame("NV12", 1280, 720, seed=3)
    name2, st2, same2 = run(other)
    print("OTHER", name2, st2[0], st2[2][:90].replace("\\n", " "), same2)
""") % (ROOT, ROOT, ROOT)


def test_a_baseline_configuration_runs_specialised_without_hiprtc(tmp_path):
    if not any(f.endswith(".co") for f in os.listdir(os.path.join(ROOT, "gyroflow_amd", "jit_cache"))) if os.path.isdir(os.path.join(ROOT, "gyroflow_amd", "jit_cache")) else True:
        pytest.skip("gyroflow_amd/jit_cache is empty (built without libhiprtc.so)")
    script = tmp_path / "s.py"
    script.write_text(SCRIPT)
    r = subprocess.run([sys.executable, str(script)], env=dict(os.environ, GFW_NO_HIPRTC="1"), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = {l.split()[0]: l for l in r.stdout.splitlines() if l.startswith(("C1", "OTHER"))}
    assert "yuv_fused_jit 2 code object from" in lines["C1"] and lines["C1"].endswith("True"), lines
    assert lines["OTHER"].split()[1] == "yuv_fused_p1" and "libhiprtc.so not found and no cached kernel" in lines["OTHER"] and lines["OTHER"].endswith("True"), lines


def test_a_compiled_kernel_is_kept_in_GFW_JIT_CACHE_and_found_by_the_next_process(tmp_path):
    cache = tmp_path / "cache"
    cache.mkdir()
    code = SCRIPT.replace('print("C1"', 'print("IGNORED"')
    script = tmp_path / "s.py"
    script.write_text(code)
    for attempt, rtc in ((1, ""), (2, "1")):                      # first process compiles and stores; the second has no hiprtc and still specialises the 720p clip
        r = subprocess.run([sys.executable, str(script)], env=dict(os.environ, GFW_JIT_CACHE=str(cache), GFW_NO_HIPRTC=rtc), capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        line = [l for l in r.stdout.splitlines() if l.startswith("OTHER")][0]
        assert line.split()[1] == "yuv_fused_p1_jit" and line.endswith("True"), (attempt, line)
        if attempt == 2:
            assert "code object from" in line, line
    assert len([f for f in os.listdir(cache) if f.endswith(".co")]) >= 1
