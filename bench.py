#!/usr/bin/env python3
"""bench.py — Mpix/s of the gfwarp hot path on N MI355X (one process per GPU).

A "step" is one pass of the hot path over one synthetic frame: all planes of a 4K (3840x2160) u16 4:2:2
frame (BASELINE.json configs[1], "C2": planar YUV422P16LE = 3 x Luma16, GoPro-style opencv_fisheye lens,
per-row rolling-shutter matrices) warped through libgfwarp's C ABI from buffers already resident in HBM.
Frames shard across ranks (weak scaling: every rank warps its own K frames; no pixel crosses GPUs); the
only collectives are a barrier and two tiny reductions (time max, checksum sum) over RCCL.

Prints ONE JSON line on rank 0 (see the driver contract in the task statement), with
  roofline     — algorithmic HBM bytes/launch (sum over planes of w*h*bpp read + written)
                 over the kernel's mean launch duration, measured with hipEvents on the launch stream
                 (GFW_OPT_PROFILE) during the timed region, against the 8 TB/s HBM3E peak;
  cpu_baseline — the oracle (C restatement of the reference CPU path, OpenMP over all host cores) on a
                 bounded sample of the same frames, rank 0 at N=1 only.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time
import zlib

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np
import torch

from gyroflow_amd import abi, shard, synthetic as S, warp

HBM_PEAK_GBS = 8000.0            # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
FMT, WIDTH, HEIGHT = "YUV422P16LE", 3840, 2160
N_DISTINCT = 4                   # host-generated source frames (the oracle's parity spot check runs on these)
N_RESIDENT = 64                  # distinct source frames + per-row matrix tables resident in HBM, cycled by the steps
                                 # (SURVEY.md 8d "64 distinct resident source frames cycled": 4.2 GB, far beyond L2 + MALL)
N_DST = 4                        # destination frame sets written round-robin


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--width", type=int, default=WIDTH)
    ap.add_argument("--height", type=int, default=HEIGHT)
    ap.add_argument("--fmt", default=FMT)
    ap.add_argument("--fov", type=float, default=1.0)
    ap.add_argument("--interp", type=int, default=2, help="2 bilinear (north-star), 4 bicubic, 8 Lanczos4 (the reference's render default, cli.rs:618)")
    ap.add_argument("--crop", action="store_true", help="C4: adaptive-zoom crop (fov 0.82 + non-zero translation2d)")
    ap.add_argument("--resident", type=int, default=N_RESIDENT, help="distinct source frames / matrix tables kept in HBM and cycled")
    ap.add_argument("--c1", action="store_true", help="C1: 1920x1080 NV12 (u8), one constant quaternion, matrix_count = 1 "
                                                      "(BASELINE.json configs[0], the reference's own CPU-runnable case)")
    ap.add_argument("--host-buffers", action="store_true",
                    help="PCIe-inclusive run: source and destination planes live in host memory (BufferSource::Cpu), every "
                         "call stages H2D, warps, copies D2H and synchronises — never the headline value")
    ap.add_argument("--profile-every", type=int, default=8,
                    help="bracket every N-th launch of the timed region with hipEvents for the roofline's kernel duration "
                         "(0 = none); the event pairs themselves cost GPU time between back-to-back kernels")
    ap.add_argument("--digital", default="", help="digital lens on top of the physical one (gopro_superview, gopro_hyperview, ...): "
                                                    "served by the fused kernel's generic-model instantiation")
    ap.add_argument("--variant", type=int, default=0)
    ap.add_argument("--grid", type=int, default=0)
    ap.add_argument("--build-matrices", action="store_true",
                    help="build every frame's per-row matrices on the device from quaternion tracks (gfw_build_matrices, "
                         "the 'next' row f-1) inside the timed region instead of using pre-packed resident tables")
    ap.add_argument("--build-batch", type=int, default=16, help="with --build-matrices: frames per gfw_build_matrices_batch call (1 = one "
                                                                 "gfw_build_matrices per frame on the auxiliary stream)")
    ap.add_argument("--upload-matrices", action="store_true",
                    help="upload the per-row matrices from host memory every frame (the reference's OpenCL backend does, "
                         "opencl.rs:406) instead of keeping the pre-packed tables of the clip resident in HBM")
    args = ap.parse_args()

    rank, local_rank, world = shard.env_rank()
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (libgfwarp has no CPU fallback)")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = shard.init("nccl", rank, world, dev)

    lib = abi.load_library()
    if lib.gfw_set_device(local_rank) != 0:
        raise SystemExit("gfw_set_device failed: %s" % lib.gfw_last_error().decode())

    # ---- synthetic clip, resident in HBM ------------------------------------------------------------
    if args.c1:
        args.width, args.height, args.fmt = 1920, 1080, "NV12"
    W, H = args.width, args.height
    readout = 0.0 if args.c1 else 16.0
    cquat = S.quat_from_euler_deg(5.0, 2.0, 3.0) if args.c1 else None
    fov = 0.82 if args.crop else args.fov
    ov = {"translation2d": (13.25, -7.5)} if args.crop else None
    lens = S.gopro_style_lens(W, H)
    if args.digital:
        lens["digital"] = args.digital
    frames = [S.SyntheticFrame(args.fmt, W, H, seed=0x9F10 + rank * 1000 + i, timestamp_ms=1000.0 + 33.3 * (rank * 1000 + i), lens=dict(lens),
                               fov=fov, base_overrides=ov, interpolation=args.interp, readout_ms=readout, constant_quat=cquat)
              for i in range(N_DISTINCT)]
    nplanes = len(frames[0].planes)
    NR = N_DISTINCT if args.host_buffers else max(N_DISTINCT, args.resident)
    base_src = [[torch.from_numpy(pl["src"]).to(dev) for pl in fr.planes] for fr in frames]
    # resident set j < 4 is host frame j itself; the others are byte-rotated copies (distinct content, same statistics)
    d_src = [[base_src[j % N_DISTINCT][p] if j < N_DISTINCT else torch.roll(base_src[j % N_DISTINCT][p], 4098 * j)
              for p in range(nplanes)] for j in range(NR)]
    # destinations start from the same fill pattern as the host copies: stride padding is never written by the warp
    d_dst = [[torch.from_numpy(pl["dst"]).to(dev) for pl in frames[0].planes] for _ in range(N_DST)]
    types = [pl["pixel_type"] for pl in frames[0].planes]
    bufsets = []                                      # bufsets[j * N_DST + d]: source set j -> destination set d
    h_dst = [[pl["dst"].copy() for pl in frames[0].planes] for _ in range(N_DST)]
    for j in range(NR):
        fr = frames[j % N_DISTINCT]
        for d in range(N_DST):
            if args.host_buffers:
                bufsets.append([warp.host_buffers(pl["src"], pl["size"], h_dst[d][p], pl["out_size"]) for p, pl in enumerate(fr.planes)])
                continue
            bufsets.append([warp.device_buffers(d_src[j][p].data_ptr(), d_src[j][p].numel(), pl["size"],
                                                d_dst[d][p].data_ptr(), d_dst[d][p].numel(), pl["out_size"])
                            for p, pl in enumerate(fr.planes)])
    # clip-invariant block (lens + per-plane KernelParams template): rank 0's copy is the one every rank uses
    blob = shard.broadcast_bytes(dist, b"".join(bytes(pl["params"]) for pl in frames[0].planes), dev)
    ksz = C.sizeof(abi.KernelParams)
    tmpl = [abi.KernelParams.from_buffer_copy(blob[k * ksz:(k + 1) * ksz]) for k in range(nplanes)]
    params = [tmpl for _ in frames]
    be = warp.Backend(params[0][0], types[0], frames[0].model, frames[0].digital, bufsets[0][0])
    stream = torch.cuda.current_stream(dev)
    be.set_stream(stream.cuda_stream)
    be.set_option(abi.OPT_SYNCHRONOUS, 1 if args.host_buffers else 0)
    if args.variant:
        be.set_option(abi.OPT_KERNEL_VARIANT, args.variant)
    if args.grid:
        be.set_option(abi.OPT_TUNE_GRID, args.grid)

    rows_n = frames[0].matrices.shape[0]

    def matrix_sets():
        """One per-row matrix table per resident frame: its own timestamp on the synthetic camera track."""
        sets = [fr.matrices for fr in frames]
        for j in range(N_DISTINCT, NR):
            sets.append(S.row_matrices(frames[0].lens, fov, (W, H), (W, H), 1000.0 + 33.3 * (rank * 1000 + j), readout,
                                       0x9F10 + rank * 1000 + j, constant_quat=cquat))
        return sets

    if args.build_matrices:
        org = S.sampled_track(11 + rank, 0.0, 1000.0 + 33.4 * (args.steps + args.warmup + 2), 1000.0)
        smo = S.sampled_track(12 + rank, 0.0, 1000.0 + 33.4 * (args.steps + args.warmup + 2), 200.0, scale=0.25)
        be.set_quaternion_tracks(org, smo)
        be.set_option(abi.OPT_MATRICES_ON_DEVICE, 2)
        nk = S.new_k(frames[0].lens, fov, W, H)
        table = be.build_matrices(nk, 1000.0, 16.0, H, H)          # context-owned table: same pointer every frame
        be.synchronize()
        calls = [warp.FrameCall(be, bufsets[j * N_DST + (j % N_DST)], params[j % N_DISTINCT], types, table, H) for j in range(NR)]
        timing = abi.FrameTiming()
        timing.frame_readout_time_ms, timing.rows, timing.readout_dim = 16.0, H, H
        for i, v in enumerate(np.asarray(nk, dtype=np.float64).reshape(9)):
            timing.new_k[i] = v
        build_fn, ctxp, tref = be.lib.gfw_build_matrices, be.ctx, C.byref(timing)
        tbl = C.c_void_p(0)
        tblref = C.byref(tbl)
        BATCH = args.build_batch
        timings = (abi.FrameTiming * max(BATCH, 1))()
        for t in timings:
            t.frame_readout_time_ms, t.rows, t.readout_dim = 16.0, H, H
            for i, v in enumerate(np.asarray(nk, dtype=np.float64).reshape(9)):
                t.new_k[i] = v
        tptrs = (C.c_void_p * max(BATCH, 1))()
        batch_fn = be.lib.gfw_build_matrices_batch
    elif args.upload_matrices:
        mats = matrix_sets()
        calls = [warp.FrameCall(be, bufsets[j * N_DST + (j % N_DST)], params[j % N_DISTINCT], types, mats[j]) for j in range(NR)]
    else:
        d_mat = [torch.from_numpy(warp.pack_matrices(m)).to(dev) for m in matrix_sets()]
        be.set_option(abi.OPT_MATRICES_ON_DEVICE, 2)
        calls = [warp.FrameCall(be, bufsets[j * N_DST + (j % N_DST)], params[j % N_DISTINCT], types, d_mat[j].data_ptr(), rows_n) for j in range(NR)]

    def step(k):
        call = calls[k % NR]
        if args.build_matrices and BATCH > 1:
            # every BATCH frames: one launch builds the tables of the next BATCH frames, in order on the warp's stream
            if k % BATCH == 0:
                for i in range(BATCH):
                    timings[i].timestamp_ms = 1000.0 + 33.3 * (k + i)
                batch_fn(ctxp, timings, BATCH, tptrs)
            call.mp = tptrs[k % BATCH]
        elif args.build_matrices:
            # the context builds this frame's table on its auxiliary stream (ring of tables) while the previous frame warps
            timing.timestamp_ms = 1000.0 + 33.3 * k
            build_fn(ctxp, tref, None, tblref)
            call.mp = tbl.value
        call()

    for k in range(args.warmup):
        step(k)
    torch.cuda.synchronize(dev)
    pe = args.profile_every
    be.set_option(abi.OPT_PROFILE, 1 if pe == 1 else 0)
    be.get_profile(reset=True)
    shard.barrier(dist)
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    if pe > 1:
        set_opt, ctxp2 = be.lib.gfw_set_option, be.ctx
        for k in range(args.steps):
            if k % pe == 0:
                set_opt(ctxp2, abi.OPT_PROFILE, 1)
                step(k)
                set_opt(ctxp2, abi.OPT_PROFILE, 0)
            else:
                step(k)
    else:
        for k in range(args.steps):
            step(k)
    t_enq = time.perf_counter() - t0                 # host time to enqueue the K steps (GPU runs behind it)
    torch.cuda.synchronize(dev)
    shard.barrier(dist)
    t1 = time.perf_counter()
    elapsed = t1 - t0
    kernel_ms, launches = be.get_profile(reset=True)
    be.set_option(abi.OPT_PROFILE, 0)

    # checksum of the last frame's planes (checksum of checksums across ranks)
    crc = 0
    last = ((args.steps - 1) % NR) % N_DST
    for p in range(nplanes):
        crc = zlib.crc32(h_dst[last][p].tobytes() if args.host_buffers else d_dst[last][p].cpu().numpy().tobytes(), crc)
    elapsed = shard.reduce_max(dist, elapsed, dev)
    rank_crcs = [g[0] for g in shard.gather_checksums(dist, [crc], dev)]
    crc = shard.reduce_checksum(dist, crc, dev)

    luma_px = frames[0].luma_pixels()
    alg_bytes = frames[0].algorithmic_bytes()
    value = luma_px * args.steps * world / elapsed / 1e6
    out = {
        "metric": "Mpix/s (4K u16 YUV, rolling-shutter warp)" if not args.c1 else "Mpix/s (1080p u8 NV12 warp)",
        "value": round(value, 2), "unit": "Mpix/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(elapsed / args.steps * 1e3, 5), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f32 coordinates, %s pixels" % np.dtype(abi.PIXEL_TYPES[types[0]][1]).name, "data": "synthetic",
        "config": {"workload": "%s: %dx%d %s, opencv_fisheye GoPro-style lens, rolling shutter matrix_count=%d, %s, "
                               "%d frames + per-row matrix tables resident in HBM%s"
                               % ("C1" if args.c1 else "C2" if (W, H, args.fmt) == (3840, 2160, FMT) else "C3" if (W, H, args.fmt) == (7680, 4320, FMT) else "C4" if args.crop else "custom",
                                  W, H, args.fmt, frames[0].matrices.shape[0],
                                  {2: "bilinear", 4: "bicubic", 8: "Lanczos4"}.get(args.interp, str(args.interp)), NR,
                                  " — HOST buffers: H2D + warp + D2H + sync per frame (PCIe-inclusive)" if args.host_buffers else
                                  " (matrices re-uploaded per frame)" if args.upload_matrices else
                                  " (per-row matrices built on the device every frame from quaternion tracks)" if args.build_matrices else ""),
                   "frames_per_rank": args.steps, "parallelism": "frame-sharded x%d" % world,
                   "backend": warp.last_backend(), "checksum": crc, "rank_checksums": rank_crcs,
                   "host_enqueue_ms_per_step": round(t_enq / args.steps * 1e3, 5)},
    }
    if launches:
        per_launch_ms = kernel_ms / launches
        achieved = alg_bytes / (per_launch_ms * 1e-3) / 1e9
        # HBM bytes per launch from the committed PMC passes of this exact workload (rocprofv3 cannot run inside bench.py)
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "r01_c2_traffic.json")
        if (W, H, args.fmt, args.interp, args.crop, args.variant, args.host_buffers) == (WIDTH, HEIGHT, FMT, 2, False, 0, False) and os.path.exists(tpath):
            tj = json.load(open(tpath))
            traffic = int((tj["fetch_size_kib"] * tj["fetch_correction"] + tj["write_size_kib"]) * 1024)
        out["roofline"] = {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                           "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
                           "kernel": warp.last_backend(), "kernel_ms_per_launch": round(per_launch_ms, 5),
                           "algorithmic_bytes_per_launch": alg_bytes, "launches": launches}
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import _oracle as O
        cores = O.lib().gfw_oracle_num_threads()
        O.run_frame(frames[0])                         # warm-up
        n_cpu = 0
        c0 = time.perf_counter()
        while True:
            ref = O.run_frame(frames[n_cpu % N_DISTINCT])
            n_cpu += 1
            if time.perf_counter() - c0 > 12.0 or n_cpu >= 256:
                break
        cpu_s = time.perf_counter() - c0
        out["cpu_baseline"] = {"value": round(luma_px * n_cpu / cpu_s / 1e6, 3), "unit": "Mpix/s", "cores": cores, "kind": "port",
                               "sample": "%d frames of the same workload through oracle/gfw_oracle.c (OpenMP rows, %d threads)" % (n_cpu, cores)}
        # parity spot-check on the frame the oracle just produced
        i = (n_cpu - 1) % N_DISTINCT
        be.set_option(abi.OPT_SYNCHRONOUS, 1)
        be.set_option(abi.OPT_MATRICES_ON_DEVICE, 0)
        be.get_profile(reset=True)
        be.undistort_frame(bufsets[i * N_DST], params[i], types, frames[i].matrices)
        ok = all(np.array_equal(ref[p], h_dst[0][p] if args.host_buffers else d_dst[0][p].cpu().numpy()) for p in range(nplanes))
        out["config"]["parity_vs_oracle"] = "bit-exact" if ok else "MISMATCH"
    be.close()
    if rank == 0:
        print(json.dumps(out), flush=True)
    shard.finish(dist)


if __name__ == "__main__":
    main()
