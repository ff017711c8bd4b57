"""RCCL on one rank: every collective the N-rank run issues (gyroflow_amd/shard.py: broadcast of the clip-invariant block, barrier, MAX all-reduce of the elapsed
time, all-gather + SUM all-reduce of the checksums, the C5 clip's ragged per-frame gather) executed through the `nccl` backend on DEVICE tensors, with the
process group bound to the device (`device_id`) exactly as bench.py binds it — so that this branch is not executed for the first time on the 8-GPU node.
Also: bench.py itself as a 1-rank RCCL job (GFW_FORCE_DIST=1 --backend nccl)."""
import json
import os
import subprocess
import sys
import textwrap

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = textwrap.dedent("""
    import os, sys
    sys.path.insert(0, %r)
    import torch
    from gyroflow_amd import shard, synthetic as S
    rank, local_rank, world = shard.env_rank()
    assert world == 1
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    dist = shard.init("nccl", rank, world, dev)
    assert dist is not None and dist.get_backend() == "nccl" and dist.get_world_size() == 1
    shard.barrier(dist)
    assert shard.reduce_max(dist, 1.25, dev) == 1.25
    assert shard.reduce_checksum(dist, 123456789, dev) == 123456789
    tmpl = bytes(S.SyntheticFrame("YUV422P16LE", 64, 32, seed=100).planes[0]["params"])
    assert shard.broadcast_bytes(dist, tmpl, dev) == tmpl and len(tmpl) == 368
    assert shard.gather_checksums(dist, [5, 7, 11], dev) == [[5, 7, 11]]
    sums, rows = shard.assemble_frame_checksums(dist, [1000 + f * f for f in range(7)], rank, world, 7, dev)
    assert sums == [1000 + f * f for f in range(7)] and rows == [sums]
    t = torch.arange(1 << 20, dtype=torch.int64, device=dev)                   # something bigger than a control word, on the device
    dist.all_reduce(t)
    assert int(t[-1].item()) == (1 << 20) - 1
    shard.barrier(dist)
    torch.cuda.synchronize()
    print("RCCL-OK", flush=True)
    shard.finish(dist)
""") % ROOT


def _env(port):
    return dict(os.environ, RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), GFW_FORCE_DIST="1", HSA_ENABLE_IPC_MODE_LEGACY="0")


def test_every_collective_of_the_sharded_run_on_device_tensors(tmp_path):
    script = tmp_path / "w.py"
    script.write_text(WORKER)
    r = subprocess.run([sys.executable, str(script)], env=_env(29541), capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "RCCL-OK" in r.stdout, (r.stdout[-500:], r.stderr[-1500:])


def test_bench_as_a_one_rank_rccl_job():
    """the driver's N > 1 launch shape with N = 1: RANK / WORLD_SIZE from the environment, process group on RCCL, control tensors on the device"""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "16", "--warmup", "4", "--backend", "nccl", "--no-cpu-baseline",
                        "--width", "1280", "--height", "720"], env=_env(29542), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    out = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert out["config"]["collectives"] == "nccl" and out["config"]["parity_vs_oracle"] == "bit-exact" and out["n_gpus"] == 1
    assert out["config"]["rank_checksums"] == [out["config"]["checksum"]]
