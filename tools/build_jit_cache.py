#!/usr/bin/env python3
"""Build step: the shipped cache of specialised kernels, gyroflow_amd/jit_cache/<hash>.co (git-ignored like every built artefact; it travels with the library).

The fused frame kernel is compiled per clip at run time (gfw_jit.hip, hiprtc) — 46 us per C2 frame against 73 for the ahead-of-time kernels.  For the BASELINE
configurations (C1-C4 as bench.py builds them, plus the common 4K formats and the LUT samplers of C2) the specialised kernels are built HERE, at build time, and
found by the library next to libgfwarp.so: a box without libhiprtc.so still runs them at the headline rate, and nobody waits for a compile.  Nothing here needs a
device: gfw_debug_jit_key derives the definition list, the bake header and the cache file name exactly as run_planes / jit_for do on one; gfw_debug_jit_compile
is the library's own hiprtc path (this container has libhiprtc.so; without it the step is skipped and the library compiles at run time as before).
usage: tools/build_jit_cache.py [--list]"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gyroflow_amd import abi, synthetic as S, warp  # noqa: E402

OUT = os.path.join(ROOT, "gyroflow_amd", "jit_cache")


def bench_frame(fmt="YUV422P16LE", w=3840, h=2160, interp=2, crop=False, c1=False, fov=1.0):
    """the frame bench.py builds for these arguments (parameters only: no pixels are generated)"""
    if c1:
        fmt, w, h = "NV12", 1920, 1080
    readout = 0.0 if c1 else 16.0
    cquat = S.quat_from_euler_deg(5.0, 2.0, 3.0) if c1 else None
    ov = {"translation2d": (13.25, -7.5)} if crop else None
    return S.SyntheticFrame(fmt, w, h, seed=0x9F10, timestamp_ms=1000.0, lens=dict(S.gopro_style_lens(w, h)), fov=0.82 if crop else fov, base_overrides=ov,
                            interpolation=interp, readout_ms=readout, constant_quat=cquat, pixels=False)


CONFIGS = [
    ("C2 4K YUV422P16LE bilinear", dict()),
    ("C1 1080p NV12 constant quaternion", dict(c1=True)),
    ("C3 8K YUV422P16LE", dict(w=7680, h=4320)),
    ("C4 4K RGBAF32 crop", dict(fmt="RGBAF32", crop=True)),
    ("C4 4K GBRAPF32LE crop", dict(fmt="GBRAPF32LE", crop=True)),
    ("C2 bicubic", dict(interp=4)),
    ("C2 Lanczos4", dict(interp=8)),
    ("4K NV12", dict(fmt="NV12")),
    ("4K P010LE", dict(fmt="P010LE")),
    ("4K YUV420P", dict(fmt="YUV420P")),
    ("4K NV12 bicubic", dict(fmt="NV12", interp=4)),
    ("4K NV12 Lanczos4", dict(fmt="NV12", interp=8)),
    ("4K P010LE bicubic", dict(fmt="P010LE", interp=4)),
]


ARCH = b"gfx950:sramecc+:xnack-"       # hipDeviceProp_t::gcnArchName of an MI355X: the string the library compiles for at run time (part of the key)


def key_of(lib, fr, arch=ARCH):
    n = len(fr.planes)
    bufs = []
    for p, pl in enumerate(fr.planes):                 # any non-null device pointers: the key holds no pointer
        bufs.append(warp.device_buffers(0x100000 * (p + 1), pl["size"][2] * pl["size"][1], pl["size"], 0x90000000 + 0x100000 * p, pl["out_size"][2] * pl["out_size"][1], pl["out_size"]))
    barr = (abi.Buffers * n)(*bufs)
    parr = (abi.KernelParams * n)(*[pl["params"] for pl in fr.planes])
    tarr = (C.c_int * n)(*[abi.PIXEL_TYPES[pl["pixel_type"]][0] for pl in fr.planes])
    defs, header, name = C.create_string_buffer(4096), C.create_string_buffer(1 << 16), C.create_string_buffer(128)
    lib.gfw_debug_jit_key.argtypes = [C.c_int, C.POINTER(abi.Buffers), C.POINTER(abi.KernelParams), C.POINTER(C.c_int), C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int,
                                      C.c_char_p, C.c_char_p, C.c_size_t, C.c_char_p, C.c_size_t, C.c_char_p, C.c_size_t]
    rc = lib.gfw_debug_jit_key(n, barr, parr, tarr, fr.model, fr.digital, None, fr.matrices.shape[0], 2, arch, defs, len(defs), header, len(header), name, len(name))
    if rc != 0:
        raise RuntimeError("gfw_debug_jit_key: %d %s" % (rc, lib.gfw_last_error().decode()))
    return defs.value, header.value, name.value.decode()


def main():
    lib = abi.load_library()
    lib.gfw_debug_jit_compile.argtypes = [C.c_char_p, C.c_char_p, C.c_char_p, C.c_char_p, C.c_char_p, C.c_size_t]
    lib.gfw_debug_jit_compile.restype = C.c_long
    os.makedirs(OUT, exist_ok=True)
    built = 0
    for label, kw in CONFIGS:
        defs, header, name = key_of(lib, bench_frame(**kw))
        path = os.path.join(OUT, name)
        if "--list" in sys.argv:
            print("%-36s %s %s" % (label, name, "present" if os.path.exists(path) else "missing"))
            continue
        if os.path.exists(path) and os.path.getsize(path) > 64 and open(path, "rb").read(4) == b"\x7fELF":
            continue                                    # (anything else under that name — a file cut short — is rebuilt; the library writes through a temporary name)
        log = C.create_string_buffer(1 << 16)
        n = lib.gfw_debug_jit_compile(ARCH, defs, header, path.encode(), log, len(log))
        if n == -2:
            sys.stderr.write("[build] libhiprtc.so not available: the shipped kernel cache stays empty (kernels are compiled at run time)\n")
            return 0
        if n <= 0:
            raise RuntimeError("%s: build failed\n%s" % (label, log.value.decode(errors="replace")[-3000:]))
        built += 1
    # kernels of other sources / options: stale entries would never be found again (the name hashes the source), so they only cost space
    keep = {key_of(lib, bench_frame(**kw))[2] for _, kw in CONFIGS}
    for f in os.listdir(OUT):
        if (f.endswith(".co") and f not in keep) or ".co.tmp" in f:
            os.remove(os.path.join(OUT, f))
    if "--list" not in sys.argv:
        print("jit_cache: %d kernels (%d built now)" % (len(keep), built))
    return 0


if __name__ == "__main__":
    sys.exit(main())
