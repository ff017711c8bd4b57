"""World-size-2 CPU (gloo) test of the multi-GPU control path used by bench.py (gyroflow_amd/shard.py)."""
import os
import subprocess
import sys
import textwrap

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = textwrap.dedent("""
    import os, sys, zlib
    sys.path.insert(0, %r)
    sys.path.insert(0, os.path.join(%r, "tests"))
    import numpy as np
    from gyroflow_amd import shard, synthetic as S
    import _oracle as O
    rank, local_rank, world = shard.env_rank()
    dist = shard.init("gloo", rank, world)
    total = 6
    mine = list(shard.frames_for_rank(rank, world, total))
    crc = 0
    for i in mine:                                   # the oracle stands in for the GPU kernel on this CPU box
        fr = S.SyntheticFrame("YUV422P16LE", 64, 32, seed=100 + i)
        for o in O.run_frame(fr, nthreads=1):
            crc = zlib.crc32(o.tobytes(), crc)
    shard.barrier(dist)
    t = shard.reduce_max(dist, 1.0 + rank)
    c = shard.reduce_checksum(dist, crc)
    # clip-invariant block: rank 0's KernelParams template reaches every rank byte-for-byte
    tmpl = S.SyntheticFrame("YUV422P16LE", 64, 32, seed=100 + rank).planes[0]["params"]      # rank-dependent on purpose
    got = shard.broadcast_bytes(dist, bytes(tmpl))
    ref0 = bytes(S.SyntheticFrame("YUV422P16LE", 64, 32, seed=100).planes[0]["params"])
    assert got == ref0 and len(got) == 368, (rank, len(got))
    g = shard.gather_checksums(dist, [crc, rank + 7])
    assert len(g) == world and g[rank] == [crc, rank + 7] and g[1 - rank][1] == (1 - rank) + 7
    assert sum(x[0] for x in g) == c
    # the C5 clip's per-frame checksums come back in clip order on every rank (bench.py --c5 uses this very function);
    # 7 frames over 2 ranks: rank 0 owns 4, rank 1 owns 3 (ragged)
    own = list(shard.frames_for_rank(rank, world, 7))
    sums, rows = shard.assemble_frame_checksums(dist, [1000 + f * f for f in own], rank, world, 7)
    assert sums == [1000 + f * f for f in range(7)], sums
    assert len(rows) == world and len(rows[0]) == 4
    try:
        shard.assemble_frame_checksums(dist, [1], rank, world, 7)
        raise SystemExit("ragged input accepted")
    except ValueError:
        pass
    print("RESULT", rank, mine, t, c, flush=True)
    shard.finish(dist)
""") % (ROOT, ROOT)


def test_two_rank_frame_sharding(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29533", OMP_NUM_THREADS="1")
    procs = []
    for r in range(2):
        e = dict(env, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE="2")
        procs.append(subprocess.Popen([sys.executable, str(script)], env=e, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    outs = [p.communicate(timeout=300)[0] for p in procs]
    res = {}
    for o in outs:
        line = [l for l in o.splitlines() if l.startswith("RESULT")]
        assert line, o
        parts = line[0].split(" ", 2)
        res[int(parts[1])] = parts[2]
    # disjoint, complete frame ownership; identical reduced values on both ranks
    assert res[0].startswith("[0, 2, 4]") and res[1].startswith("[1, 3, 5]")
    assert res[0].split("]")[1] == res[1].split("]")[1]
    assert " 2.0 " in res[0] + " "


def test_frames_for_rank_partition():
    from gyroflow_amd import shard
    for world in (1, 2, 4, 8):
        seen = sorted(i for r in range(world) for i in shard.frames_for_rank(r, world, 10000))
        assert seen == list(range(10000))
