"""The torch (device-capable) frame generator is byte-identical to the numpy one; bench.py's launcher reports a worker's
own error without re-running it."""
import os
import subprocess
import sys

import numpy as np
import pytest

from gyroflow_amd import synthetic as S

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("ptype,w,h,mv", [("Luma16", 130, 70, None), ("Luma16", 64, 33, 1023.0), ("Luma8", 100, 40, None),
                                           ("UV16", 33, 20, None), ("UV8", 64, 16, None), ("RGBAf", 40, 30, None),
                                           ("R32f", 50, 20, 255.0), ("RGB8", 31, 17, None), ("RGBA16", 20, 9, None)])
def test_torch_pattern_equals_numpy_pattern(ptype, w, h, mv):
    a, st = S.make_plane_buffer(w, h, ptype, 0x9F10 + 3, mv)
    b, st2 = S.make_plane_buffer_torch(w, h, ptype, 0x9F10 + 3, mv)
    assert st == st2 and np.array_equal(a, b.numpy())


def test_frame_without_host_pixels_has_the_same_geometry():
    a = S.SyntheticFrame("YUV422P16LE", 130, 66, seed=4)
    b = S.SyntheticFrame("YUV422P16LE", 130, 66, seed=4, pixels=False)
    assert np.array_equal(a.matrices, b.matrices)
    for pa, pb, dev in zip(a.planes, b.planes, b.device_planes("cpu")):
        assert pa["size"] == pb["size"] and pa["out_size"] == pb["out_size"] and bytes(pa["params"]) == bytes(pb["params"])
        assert pb["src"] is None and np.array_equal(pa["src"], dev.numpy())
    for pa, o in zip(a.planes, b.device_outputs("cpu")):
        assert np.array_equal(pa["dst"], o.numpy())


def test_fast_track_sampler_matches_the_scalar_one():
    ts, q = S.sampled_track_fast(11, 0.0, 200.0, 500.0, scale=0.25)
    ts2, q2 = S.sampled_track(11, 0.0, 200.0, 500.0, scale=0.25)
    assert np.array_equal(ts, ts2) and np.abs(q - q2).max() < 1e-14


def test_bench_launcher_without_a_gpu_fails_loudly_and_does_not_retry():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1"],
                       capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 1 and r.stdout.strip() == ""
    assert r.stderr.count("attempt") == 1 and "needs a GPU" in r.stderr
