"""Frame sharding across ranks (one process per GPU) — the control path of the multi-GPU run.

Frames of a clip are independent once clip-level state is fixed (frame_transform.rs:165), so the data path has no
collective: rank r owns a contiguous-by-stride set of frame indices and its own resident buffers.  The only
communication is one broadcast of the clip-invariant parameter block, a barrier either side of the timed region, a MAX
reduction of the elapsed time, and an all-gather / SUM reduction of output checksums (checksum of checksums).  Backend "nccl" is RCCL over xGMI on ROCm; tests run it on "gloo".
"""
import os

import torch


def env_rank():
    return int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))


def frames_for_rank(rank, world, total_frames):
    """Round-robin frame ownership: rank r handles frames r, r+world, ...."""
    return range(rank, total_frames, world)


def init(backend, rank, world, device=None):
    # GFW_FORCE_DIST=1 initialises the process group even for a single rank (lets a 1-GPU box exercise the RCCL calls)
    if world <= 1 and os.environ.get("GFW_FORCE_DIST", "") == "":
        return None
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29511")
    kw = {}
    if backend == "nccl" and device is not None:
        kw["device_id"] = device
    dist.init_process_group(backend=backend, rank=rank, world_size=world, **kw)
    return dist


def barrier(dist):
    if dist is not None:
        dist.barrier()


def reduce_max(dist, value, device="cpu"):
    if dist is None:
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def reduce_checksum(dist, crc, device="cpu"):
    """Order-independent combination of the per-rank CRC32s: sum modulo 2^63."""
    if dist is None:
        return int(crc)
    t = torch.tensor([int(crc)], dtype=torch.int64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return int(t.item())


def broadcast_bytes(dist, payload, device="cpu", src=0):
    """Rank `src` hands the clip-invariant block (lens + KernelParams template, a few hundred bytes) to every rank
    (one broadcast, rank 0 -> all).  Every rank passes a payload of the same length."""
    if dist is None:
        return bytes(payload)
    t = torch.frombuffer(bytearray(payload), dtype=torch.uint8).to(device)
    dist.broadcast(t, src=src)
    return bytes(t.cpu().numpy().tobytes())


def gather_checksums(dist, crcs, device="cpu"):
    """All-gather of the per-frame output checksums (8 B per frame): returns [world][len(crcs)] as lists of ints.
    Every rank contributes the same number of frames (weak scaling)."""
    if dist is None:
        return [[int(c) for c in crcs]]
    t = torch.tensor([int(c) for c in crcs], dtype=torch.int64, device=device)
    outs = [torch.empty_like(t) for _ in range(dist.get_world_size())]
    dist.all_gather(outs, t)
    return [[int(v) for v in o.cpu().tolist()] for o in outs]


def assemble_frame_checksums(dist, mine, rank, world, total_frames, device="cpu"):
    """Per-frame checksums of a round-robin sharded clip, back in clip order on every rank.

    ``mine[i]`` is the checksum of the i-th frame this rank owns (``frames_for_rank(rank, world, total_frames)[i]``).
    Ranks own ceil or floor(total/world) frames; the all-gather carries ceil(total/world) values per rank (8 B each,
    80 KB for the 10 000-frame clip).  Returns (frame_sums: list of total_frames ints, per_rank: the gathered rows)."""
    per_rank = (total_frames + world - 1) // world
    own = frames_for_rank(rank, world, total_frames)
    if len(mine) != len(own):
        raise ValueError("rank %d owns %d frames but reported %d checksums" % (rank, len(own), len(mine)))
    padded = [int(v) for v in mine] + [0] * (per_rank - len(mine))
    gathered = gather_checksums(dist, padded, device)
    frame_sums = [0] * total_frames
    for r in range(world):
        for i, f in enumerate(frames_for_rank(r, world, total_frames)):
            frame_sums[f] = gathered[r][i]
    return frame_sums, gathered


def finish(dist):
    if dist is not None:
        dist.destroy_process_group()
