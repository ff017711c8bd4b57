"""bench.py under the driver's own command lines (subprocess, fresh process each): it must exit 0, print ONE JSON line
carrying `roofline`, `cpu_baseline` and a bit-exact parity verdict on frames of the timed region — and the N-rank
paths (self-spawned ranks, torchrun-style environment, the C5 clip) must agree with the 1-rank run."""
import json
import os
import subprocess
import sys
import time

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run_bench(argv, env=None, timeout=600):
    e = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
        e.pop(k, None)
    e.update(env or {})
    t0 = time.time()
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + argv, env=e, capture_output=True, text=True, timeout=timeout, cwd=ROOT)
    wall = time.time() - t0
    assert r.returncode == 0, "bench.py %s -> rc %d\n%s\n%s" % (argv, r.returncode, r.stdout[-2000:], r.stderr[-4000:])
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout
    return json.loads(lines[0]), wall


def test_driver_command_line():
    out, wall = run_bench(["--gpus", "1", "--steps", "20", "--warmup", "5"])
    assert out["n_gpus"] == 1 and out["steps"] == 20 and out["warmup"] == 5
    assert out["unit"] == "Mpix/s" and out["value"] > 0 and out["higher_is_better"] is True and out["scaling"] == "weak"
    assert out["config"]["workload"].startswith("C2:")
    assert out["config"]["parity_vs_oracle"] == "bit-exact", out["config"]
    # frame 0 through the timed region's call path = what the reference's own kernel wrote for it (tests/golden/ref_golden.json)
    assert out["config"]["parity_vs_reference_kernel"].startswith("bit-exact"), out["config"]["parity_vs_reference_kernel"]
    rf = out["roofline"]
    assert rf["bound"] == "hbm" and rf["peak"] == 8000.0 and rf["launches"] >= 1
    assert abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-3
    # the steps leave as clip launches of the run-time specialised kernel (20 steps = two calls of 10 frames, no runt launch; every other launch bracketed)
    assert out["config"]["jit"]["state"] == "ready" and out["config"]["backend"].endswith("_jit"), out["config"]
    assert out["config"]["clip_frames_per_call"] == 10 and rf["frames_per_launch"] == 10.0, rf
    assert rf["algorithmic_bytes_per_launch"] == 66355200 * 10
    assert abs(rf["kernel_ms_per_frame"] * rf["frames_per_launch"] - rf["kernel_ms_per_launch"]) < 1e-3
    assert out["config"]["preheat_ms"] >= 50.0
    cb = out["cpu_baseline"]
    assert cb["kind"] == "port" and cb["cores"] >= 1 and cb["value"] > 0
    assert out["ms_per_step"] * out["steps"] / 1e3 < wall
    assert out["launcher"]["attempts"] == 1 and out["launcher"]["failures"] == []
    # the value is consistent with the step time
    assert abs(out["value"] - 3840 * 2160 / (out["ms_per_step"] * 1e-3) / 1e6) / out["value"] < 1e-3


def test_ahead_of_time_kernels_frame_by_frame():
    out, _ = run_bench(["--gpus", "1", "--steps", "10", "--warmup", "3", "--no-cpu-baseline", "--jit", "0", "--clip", "1"])
    assert out["config"]["backend"] == "yuv_fused_p1" and out["config"]["jit"]["state"] == "none"
    assert out["config"]["parity_vs_oracle"] == "bit-exact"
    assert out["config"]["parity_vs_reference_kernel"].startswith("bit-exact"), out["config"]["parity_vs_reference_kernel"]
    assert out["roofline"]["algorithmic_bytes_per_launch"] == 66355200 and out["roofline"]["frames_per_launch"] == 1.0


def test_torchrun_style_environment_single_rank():
    out, _ = run_bench(["--gpus", "1", "--steps", "8", "--warmup", "2", "--no-cpu-baseline"],
                       env={"RANK": "0", "LOCAL_RANK": "0", "WORLD_SIZE": "1", "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": "29547"})
    assert out["n_gpus"] == 1 and "launcher" not in out
    assert out["config"]["parity_vs_oracle"] == "bit-exact"


def test_two_self_spawned_ranks_on_one_gpu():
    # the N-rank control path on real hardware: both ranks share cuda:0, collectives over gloo (RCCL refuses two ranks on one device)
    one, _ = run_bench(["--gpus", "1", "--steps", "6", "--warmup", "2", "--no-cpu-baseline", "--resident", "8"])
    two, _ = run_bench(["--gpus", "2", "--steps", "6", "--warmup", "2", "--no-cpu-baseline", "--resident", "8", "--backend", "gloo", "--same-device"])
    assert two["n_gpus"] == 2 and two["launcher"]["ranks_spawned"] == 2
    assert two["config"]["frames_total"] == 12 and len(two["config"]["rank_checksums"]) == 2
    assert two["config"]["parity_vs_oracle"] == "bit-exact"
    # rank 0 of the 2-rank run warps the same frames as the 1-rank run
    assert two["config"]["rank_checksums"][0] == one["config"]["rank_checksums"][0]
    assert two["config"]["rank_checksums"][1] != two["config"]["rank_checksums"][0]


def test_eight_self_spawned_ranks_on_one_gpu():
    """The launcher path the driver's 8-GPU scaling run takes, with every rank on cuda:0 (small frames, gloo): eight workers rendezvous, broadcast the clip-invariant
    block, time the same region and all-gather their checksums — the first 8-GPU run is not also the first 8-process run."""
    out, _ = run_bench(["--gpus", "8", "--steps", "4", "--warmup", "1", "--no-cpu-baseline", "--resident", "4", "--width", "640", "--height", "360",
                        "--backend", "gloo", "--same-device"], timeout=900)
    assert out["n_gpus"] == 8 and out["launcher"]["ranks_spawned"] == 8 and out["launcher"]["failures"] == []
    assert out["config"]["frames_total"] == 32 and len(out["config"]["rank_checksums"]) == 8
    assert len(set(out["config"]["rank_checksums"])) == 8              # every rank warped its own frames
    assert out["config"]["parity_vs_oracle"] == "bit-exact"


def test_c5_clip_checksums_do_not_depend_on_the_rank_count():
    base = ["--c5", "--frames", "70", "--resident", "8", "--warmup", "3", "--no-cpu-baseline"]
    one, _ = run_bench(base + ["--gpus", "1"])
    two, _ = run_bench(base + ["--gpus", "2", "--backend", "gloo", "--same-device"])
    for o in (one, two):
        assert o["scaling"] == "strong" and o["config"]["workload"].startswith("C5:") and o["config"]["frames_total"] == 70
        assert o["config"]["parity_vs_oracle"] == "bit-exact", o["config"]
    assert one["steps"] == 70 and two["steps"] == 35
    assert one["config"]["checksum"] == two["config"]["checksum"]


@pytest.mark.parametrize("words", [1, 2, 3, 511, 2048 * 256 * 2 * 4 + 6, 2048 * 256 * 2 * 5 - 1, 3840 * 2160 * 4 // 8])
def test_checksum64_is_the_sum_of_the_words(words):
    """gfw_checksum64 (the per-frame checksum of the C5 clip): the little-endian u64 words summed modulo 2^64, accumulated into *out — through the
    unrolled main loop, its remainder loop and the odd tail word."""
    import numpy as np
    import torch
    from gyroflow_amd import abi
    lib = abi.load_library()
    assert lib.gfw_set_device(0) == 0
    rng = np.random.default_rng(words)
    host = rng.integers(0, 2 ** 64, size=words, dtype=np.uint64)
    dev = torch.device("cuda", 0)
    buf = torch.from_numpy(host.view(np.int64)).to(dev)
    out = torch.full((1,), 5, dtype=torch.int64, device=dev)
    torch.cuda.synchronize(dev)
    fr = __import__("gyroflow_amd.synthetic", fromlist=["SyntheticFrame"]).SyntheticFrame("NV12", 64, 32, seed=1)
    from gyroflow_amd import warp
    pl = fr.planes[0]
    be = warp.Backend(pl["params"], pl["pixel_type"], fr.model, fr.digital, warp.host_buffers(pl["src"], pl["size"], pl["dst"].copy(), pl["out_size"]))
    try:
        assert lib.gfw_checksum64(be.ctx, buf.data_ptr(), words * 8, out.data_ptr()) == 0
        be.synchronize()
    finally:
        be.close()
    want = (int(host.sum(dtype=np.uint64)) + 5) % (1 << 64)
    assert int(out.cpu().numpy().view(np.uint64)[0]) == want
