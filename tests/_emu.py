"""TEST INFRASTRUCTURE: the fused kernel's SOURCE, interpreted on the host — so that the CPU tier holds the product's device code itself
(not only the oracle) to the reference's numbers without a GPU.

What is built: the text libgfwarp.so embeds for hiprtc (tools/gen_jit_source.py: gfw_frame.hip + its headers), unedited except that its seven
inline-asm statements (AMD mnemonics; two of them empty optimisation barriers) become calls of same-named C functions; in front of it tests/emu/emu_prelude.h (what hiprtc's implicit
HIP environment provides: vector types, threadIdx, __shared__, the handful of amdgcn builtins the source uses) and the clip's bake header
(tests/_bake.py, the test-side restatement of gfw_api.hip's); behind it tests/emu/emu_driver.inc (the launch, lane by lane: 256 cooperative
fibers per workgroup, rendezvous at __syncthreads and at wavefront fences).  Compiled as C++17 for x86-64 with -ffp-contract=off, like the
device build.  The host side here restates what gfw_api.hip prepares for a launch (template arguments as jit_for picks them, the first
pass's s(rho) table and certificate half-width as p1_setup computes them).

It is not a product path and cannot become one: nothing under gyroflow_amd/ imports it, the library does not know it, a 640x360 frame takes a
second.  What it buys: tests/test_emu_kernel.py runs the kernel's own arithmetic, tile walk, first-pass certificate, LDS queue and address
computations against the oracle and against the reference kernel's fixture on every CPU-only run, and kernel changes can be checked for
parity before any GPU time is spent on measuring them."""
import ctypes as C
import hashlib
import math
import os
import re
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import gen_jit_source as G  # noqa: E402

from gyroflow_amd import abi, warp  # noqa: E402
import _bake  # noqa: E402

EMU = os.path.join(ROOT, "tests", "emu")
OUT = os.path.join(ROOT, "build", "emu")
CXX = "/opt/rocm/lib/llvm/bin/clang++" if os.path.exists("/opt/rocm/lib/llvm/bin/clang++") else "g++"
P1_TABLE_N = 8192                 # GFW_P1_TABLE_N
GUARD = False                     # memory-safety runs ("end" / "start"): the frame's own plane buffers as they are (the caller placed them against guard
                                  # pages), no padding; matrix tables and the first-pass table go against guard pages here
_guard_keep = []


def guarded(data, at_end):
    """a copy of `data` (any contiguous array) whose last (at_end) or first byte borders an inaccessible page"""
    import mmap
    raw = np.ascontiguousarray(data).view(np.uint8).reshape(-1)
    page = mmap.PAGESIZE
    body = (len(raw) + page - 1) // page * page
    mm = mmap.mmap(-1, body + 2 * page)
    base = C.addressof(C.c_char.from_buffer(mm))
    libc = C.CDLL("libc.so.6")
    libc.mprotect.argtypes = [C.c_void_p, C.c_size_t, C.c_int]
    assert libc.mprotect(base, page, 0) == 0 and libc.mprotect(base + page + body, page, 0) == 0
    arr = np.frombuffer(mm, dtype=np.uint8, count=len(raw), offset=page + (body - len(raw) if at_end else 0))
    arr[:] = raw
    _guard_keep.append(mm)
    return arr


def kernel_source(top="gfw_frame.hip", n_asm=13):
    """`top` + the project headers it includes as one text (gen_jit_source.expand, without its typedef prelude: the host has <stdint.h>), asm -> emu_*()."""
    out = []
    G.expand(os.path.join(G.CSRC, top), set(), out)
    src = "".join(out)

    def fix(m):
        mm = re.match(r'asm\("(\w*)[^"]*"\s*:\s*"[=+]v"\((\w+)\)(?:\s*:\s*(.*))?\);', m.group(0))
        assert mm, m.group(0)
        op, dst, ins = mm.group(1), mm.group(2), mm.group(3)
        if not op:
            return "/* asm(\"\"): an optimisation barrier */;"
        return "%s = emu_%s(%s);" % (dst, op, ", ".join(re.findall(r'"\w"\(([^)]+)\)', ins)))
    src, n = re.subn(r'asm\(.*\);', fix, src)
    assert n == n_asm, "inline-asm statements in %s: %d (the emulator knows %d)" % (top, n, n_asm)
    return src


def build(defs, header, top="gfw_frame.hip", n_asm=13, driver="emu_driver.inc", extra_flags=("-DGFW_JIT=1", "-DGFW_BAKE=1"), opt="-O1"):
    """-> path of the host library for these template arguments (+ bake header) (cached under build/emu/ by content)."""
    if not CXX.endswith("clang++"):
        import pytest
        pytest.skip("the host interpreter is built with the ROCm clang++ (half-precision and vector extensions of the kernel headers); not found")
    os.makedirs(OUT, exist_ok=True)
    text = ('#include "emu_prelude.h"\n' + header + "\n" + kernel_source(top, n_asm) + '\n#include "%s"\n' % driver)
    flags = ["-std=c++17", opt, "-ffp-contract=off", "-fno-fast-math", "-fPIC", "-shared", "-Wno-everything", "-I" + EMU] + list(extra_flags) + \
            ["-D%s=%s" % kv for kv in sorted(defs.items())]
    key = hashlib.sha256((text + " ".join(flags) + "".join(open(os.path.join(EMU, f)).read() for f in sorted(os.listdir(EMU)))).encode()).hexdigest()[:20]
    so = os.path.join(OUT, "emu_%s.so" % key)
    if not os.path.exists(so):
        # names of this process's own: two pytest-xdist workers may build the same key at once, and a source file rewritten under a running compiler ends it with SIGBUS
        cpp = os.path.join(OUT, "emu_%s.%d.cpp" % (key, os.getpid()))
        tmp = "%s.%d.tmp" % (so, os.getpid())
        open(cpp, "w").write(text)
        r = subprocess.run([CXX] + flags + [cpp, "-o", tmp, "-lm"], capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("emulator build failed:\n" + r.stderr[-4000:])
        if os.environ.get("GFW_EMU_KEEP_CPP") == "1":
            import shutil
            shutil.copyfile(cpp, os.path.join(OUT, "emu_%s.cpp" % key))
        os.remove(cpp)                                    # (the generated translation unit — half a megabyte of text per configuration — is not kept: GFW_EMU_KEEP_CPP=1 for debugging)
        os.replace(tmp, so)
    return so


def _A_derivs(rho):
    """A(rho) = atan(sqrt rho) / sqrt rho and its first two derivatives (gfw_api.hip p1_A_derivs)"""
    if rho < 1.0 / 64.0:
        A = A1 = A2 = 0.0
        pw = 1.0
        for n in range(14):
            A += pw / (2 * n + 1)
            pw *= -rho
        p1 = 1.0
        for n in range(1, 14):
            A1 += -float(n) / (2 * n + 1) * p1
            p1 *= -rho
        p2 = 1.0
        for n in range(2, 14):
            A2 += float(n) * (n - 1) / (2 * n + 1) * p2
            p2 *= -rho
        return A, A1, A2
    r = math.sqrt(rho)
    A = math.atan(r) / r
    A1 = (1.0 / (1.0 + rho) - A) / (2.0 * rho)
    A2 = (-1.0 / ((1.0 + rho) * (1.0 + rho)) - 3.0 * A1) / (2.0 * rho)
    return A, A1, A2


def s_derivs(rho, k):
    """s(rho) = A P(w), w = rho A^2, and its first two derivatives (gfw_api.hip p1_s_derivs)"""
    A, A1, A2 = _A_derivs(rho)
    w, w1, w2 = rho * A * A, A / (1.0 + rho), A1 / (1.0 + rho) - A / ((1.0 + rho) * (1.0 + rho))
    P = 1.0 + w * (k[0] + w * (k[1] + w * (k[2] + w * k[3])))
    P1 = k[0] + w * (2.0 * k[1] + w * (3.0 * k[2] + w * 4.0 * k[3]))
    P2 = 2.0 * k[1] + w * (6.0 * k[2] + w * 12.0 * k[3])
    return A * P, A1 * P + A * P1 * w1, A2 * P + 2.0 * A1 * P1 * w1 + A * (P2 * w1 * w1 + P1 * w2)


def p1_table(p0, matrices, matrix_count):
    """gfw_api.hip p1_setup / p1_prepare_table restated: the s(rho) table of the certified first pass (f64 -> float2 entries), its range and
    the certificate's coefficients.  -> (table [N+1][2] f32, rho_max, rho_scale, (e0, ew, em, host's E)) or None when the certified pass is not used."""
    k = [float(p0.k[i]) for i in range(4)]
    hrs = bool(p0.flags & abi.FLAG_HORIZONTAL_RS)
    m = np.asarray(matrices, dtype=np.float64)[matrix_count >> 1]
    rho_max = 0.0
    for y in (0.0, p0.output_height * 0.5, float(p0.output_height)):
        for x in (0.0, p0.output_width * 0.5, float(p0.output_width)):
            ox, oy = x + p0.translation2d[0], y + p0.translation2d[1]
            X, Y, Wd = ox * m[0] + oy * m[1] + m[2], ox * m[3] + oy * m[4] + m[5], ox * m[6] + oy * m[7] + m[8]
            rho_max = 1e9 if not Wd > 0.05 else max(rho_max, (X * X + Y * Y) / (Wd * Wd))
    rho_max = min(min(rho_max * 1.25 + 0.01, 64.0) * 1.15, 64.0)
    rho_max = float(np.float32(rho_max))

    def s_of(rho):
        if all(v == 0.0 for v in k):
            return 1.0
        r = math.sqrt(rho)
        if r == 0.0:
            return 1.0
        t = math.atan(r)
        t2 = t * t
        return t * (1.0 + t2 * (k[0] + t2 * (k[1] + t2 * (k[2] + t2 * k[3])))) / r
    n = P1_TABLE_N
    h = rho_max / n
    s = [s_of(i * h) for i in range(n + 1)]
    tab = np.zeros((n + 1, 2), dtype=np.float32)
    for i in range(n):
        tab[i] = (s[i], s[i + 1] - s[i])
    tab[n] = (s[n], 0.0)
    # derived bounds on s and its derivatives over [0, rho_max] (gfw_api.hip p1_prepare_table: the comment there has the derivation)
    smax, slope, s2max, u1, u2, t32, kappa = 1.0, 0.0, 0.0, 0.0, 0.0, 0.0, 3.0
    if any(v != 0.0 for v in k):
        wm = math.atan(math.sqrt(rho_max)) ** 2
        a0, a1, a2, a3 = (abs(v) for v in k)
        P0 = 1.0 + wm * (a0 + wm * (a1 + wm * (a2 + wm * a3)))
        Pd1 = a0 + wm * (2.0 * a1 + wm * (3.0 * a2 + wm * 4.0 * a3))
        Pd2 = 2.0 * a1 + wm * (6.0 * a2 + wm * 12.0 * a3)
        Pd3 = 6.0 * a2 + wm * 24.0 * a3
        L1 = P0 / 3.0 + Pd1
        L2 = 0.4 * P0 + 2.0 * Pd1 + Pd2
        L3 = (6.0 / 7.0) * P0 + 1.2 * Pd1 + (Pd2 + Pd1 * 4.0 / 3.0) + (Pd3 + 4.0 * Pd2 + Pd1 * 46.0 / 15.0)
        ng = 4096
        g = rho_max / ng
        smax = 0.0
        for i in range(ng):
            ds, ds1, ds2 = s_derivs(i * g, k)
            hi = (i + 1) * g
            b0, b1, b2 = abs(ds) + g * L1, abs(ds1) + g * L2, abs(ds2) + g * L3
            smax, slope, s2max = max(smax, b0), max(slope, b1), max(s2max, b2)
            u1, u2, t32 = max(u1, math.sqrt(hi) * b1), max(u2, hi * b1), max(t32, hi * math.sqrt(hi) * b2)
        slack = 1.0 + 1e-9
        smax, slope, s2max, u1, u2, t32 = smax * slack, slope * slack, s2max * slack, u1 * slack, u2 * slack, t32 * slack
        rp, kp = 0.0, 1.0
        gw = wm / 4096.0
        for i in range(4097):
            t2 = gw * i
            t4 = t2 * t2
            t6, t8 = t4 * t2, t4 * t4
            P = abs(1.0 + k[0] * t2 + k[1] * t4 + k[2] * t6 + k[3] * t8) - gw * Pd1
            t2h = t2 + gw
            t4h = t2h * t2h
            t6h, t8h = t4h * t2h, t4h * t4h
            Pabs = 1.0 + a0 * t2h + a1 * t4h + a2 * t6h + a3 * t8h
            dPabs = 2.0 * a0 * t2h + 4.0 * a1 * t4h + 6.0 * a2 * t6h + 8.0 * a3 * t8h
            if not P > 1e-3:
                rp = kp = 1e30
                break
            rp, kp = max(rp, dPabs / P), max(kp, Pabs / P)
        kappa = 4.5 * rp + 4.0 * kp + 6.5
    etab = h * h / 8.0 * s2max + (smax + h * slope) / 16777216.0
    f = abs(float(p0.f[0] if hrs else p0.f[1]))
    c = abs(float(p0.c[0] if hrs else p0.c[1]))
    rmax = math.sqrt(rho_max)
    vmag = f * rmax * smax + c
    u24 = 1.05 / 16777216.0
    G = f * smax + 2.0 * math.sqrt(2.0) * f * rmax * rmax * slope
    e0 = u24 * (G * 10.0 * rmax + 6.0 * f * rmax * rho_max * slope + f * rmax * smax * (kappa + 4.0) + 2.0 * vmag) + 2.0 * f * rmax * etab + 1.0 / 16384.0
    ew, em = u24 * G * rmax, u24 * G
    # the host's view of omega, mu (the kernel's own evaluation decides; this one only whether the certified instantiation is launched)
    x0, y0 = float(p0.translation2d[0]), float(p0.translation2d[1])
    x1, y1 = x0 + p0.output_width, y0 + p0.output_height
    ax, ay = max(abs(x0), abs(x1)), max(abs(y0), abs(y1))
    px, py, pw = ax * abs(m[0]) + ay * abs(m[1]), ax * abs(m[3]) + ay * abs(m[4]), ax * abs(m[6]) + ay * abs(m[7])
    wden = max(m[8] - pw, 1.0 / 1024.0, (pw + abs(m[8])) / 8.0)
    eps = e0 + ew * 3.0 * pw / wden + em * 3.0 * max(px, py) / wden
    if matrix_count <= 1 or not eps < 0.2:
        return None
    lat = np.array([smax * (1.0 + 1e-6), u1 * (1.0 + 1e-6), u2 * (1.0 + 1e-6), t32 * (1.0 + 1e-6), 4.0 * u24 * vmag + 1.0 / 131072.0, 0.0], dtype=np.float32)
    return tab, rho_max, float(np.float32(n / rho_max)), (float(np.float32(e0)), float(np.float32(ew)), float(np.float32(em)), float(eps)), lat


SAMPLE_KIND = {"Luma8": (1, 1), "Luma16": (2, 1), "RGB8": (1, 3), "RGBA8": (1, 4), "BGRA8": (1, 4), "RGB16": (2, 3), "RGBA16": (2, 4), "AYUV16": (2, 4),
               "RGBAf": (4, 4), "R32f": (4, 1), "RGBAf16": (3, 4)}          # plane 0's sample kind (bytes; 3 = two-byte float) and channel count (build_yuv_args); UV planes first: per-plane kernel


RADIAL_TABLE_MODELS = {abi.MODELS[m] for m in ("gopro", "sony", "generic_polynomial", "poly3", "poly5", "ptlens")}      # gfw_api_certificate.inc: p1_model_radial


def p1_table_radial(fr):
    """The certified first pass of a clip under a radial lens model other than the fisheye (GoPro, Sony, generic polynomial, poly3 / poly5 / ptlens) as the library sets it up (gfw_api_certificate.inc: p1_setup_radial): the table over r comes from the LIBRARY's own
    host code (gfw_debug_p1_radial — the derivation is long and this is its only statement; what the interpreter adds is the audit of every certificate it issues),
    the coefficients of E are restated here.  -> (table, rho_max, rho_scale, (e0, ew, em, E), lat) like p1_table, or None."""
    p0 = fr.planes[0]["params"]
    mc = p0.matrix_count
    if mc <= 1:
        return None
    hrs = bool(p0.flags & abi.FLAG_HORIZONTAL_RS)
    m = np.asarray(fr.matrices, dtype=np.float64)[mc >> 1]
    rho_max = 0.0
    for y in (0.0, p0.output_height * 0.5, float(p0.output_height)):
        for x in (0.0, p0.output_width * 0.5, float(p0.output_width)):
            ox, oy = x + p0.translation2d[0], y + p0.translation2d[1]
            X, Y, W = ox * m[0] + oy * m[1] + m[2], ox * m[3] + oy * m[4] + m[5], ox * m[6] + oy * m[7] + m[8]
            rho_max = 1e9 if not W > 0.05 else max(rho_max, (X * X + Y * Y) / (W * W))
    rho_max = min(rho_max * 1.25 + 0.01, 64.0)
    r_need = min(math.sqrt(rho_max) * 1.08, 8.0)
    lib = abi.load_library()
    lib.gfw_debug_p1_radial.argtypes = [C.c_void_p, C.c_int, C.c_double, C.c_void_p, C.c_void_p]
    tab, out = np.zeros((8193, 2), np.float32), np.zeros(7, np.float64)
    if lib.gfw_debug_p1_radial(C.byref(p0), fr.model, r_need, tab.ctypes.data, out.ctypes.data) != 1:
        return None
    rmax, Tmax, T1, T2, etab, nu2, _ = out
    f, cc = abs(float(p0.f[0] if hrs else p0.f[1])), abs(float(p0.c[0] if hrs else p0.c[1]))
    u24, vmag = 1.05 / 16777216.0, f * rmax * Tmax + abs(cc)
    G = f * (Tmax + 2.0 * rmax * T1)
    e0 = u24 * (G * 10.0 * rmax + 6.0 * f * rmax * rmax * T1 + 11.0 * vmag) + 2.0 * f * rmax * etab + 1.05 * f * nu2 + 1.0 / 16384.0
    ew, em = u24 * G * rmax, u24 * G
    x0, y0 = p0.translation2d[0], p0.translation2d[1]
    ax, ay = max(abs(x0), abs(x0 + p0.output_width)), max(abs(y0), abs(y0 + p0.output_height))
    px, py, pw = ax * abs(m[0]) + ay * abs(m[1]), ax * abs(m[3]) + ay * abs(m[4]), ax * abs(m[6]) + ay * abs(m[7])
    wden = max(m[8] - pw, max(1.0 / 1024.0, (pw + abs(m[8])) / 8.0))
    eps = e0 + ew * 3.0 * pw / wden + em * 3.0 * max(px, py) / wden
    if not eps < 0.2:
        return None
    lat = np.array([Tmax * (1 + 1e-6), 0.5 * T1 * (1 + 1e-6), 0.5 * rmax * T1 * (1 + 1e-6), 0.25 * (rmax * T2 + T1) * (1 + 1e-6), 4.0 * u24 * vmag + 1.0 / 131072.0, 1.0], np.float32)
    # (the driver derives the table's last key from rho_max in the r form: sqrtf)
    return tab, np.float32(rmax * rmax * (1.0 - 1e-6)), np.float32(8192.0 / rmax), (np.float32(e0), np.float32(ew), np.float32(em), eps), lat


def launch_shape(fr):
    """Template arguments of the frame's instantiation, as gfw_api.hip's build_yuv_args + jit_for choose them (fisheye, no extras)."""
    pls = fr.planes
    p0 = pls[0]["params"]
    t0 = pls[0]["pixel_type"]
    bps, n0 = SAMPLE_KIND[t0]
    il = len(pls) == 2 and pls[1]["pixel_type"] in ("UV8", "UV16")
    dw = dh = 1
    if len(pls) >= 2:
        dw, dh = p0.output_width // pls[1]["out_size"][0], p0.output_height // pls[1]["out_size"][1]
    return bps, n0, dw, dh, il


class Common(C.Structure):
    """GfwCommon (gfw_warp.h): lens ids + the host-libm-evaluated uniforms of the generic-model bodies"""
    _fields_ = [("matrices", C.c_void_p), ("mesh", C.c_void_p), ("mesh_len", C.c_int32), ("model", C.c_int32), ("digital", C.c_int32), ("pad_", C.c_int32),
                ("rot_cos", C.c_float), ("rot_sin", C.c_float), ("frame_w", C.c_float), ("frame_h", C.c_float), ("gopro_tt", C.c_float), ("pad2_", C.c_float)]


def feature_bits(fr, mesh=None):
    """`extras` as gfw_api.hip's build_yuv_args derives it: 1 IBIS/OIS terms in the matrix rows, 2 digital lens, 4 refraction, 8 lens-correction
    blend, 16 background mode 3, 32 Sony mesh / focal-plane distortion"""
    p0 = fr.planes[0]["params"]
    e = 0
    if mesh is not None and len(mesh):
        e |= 32
    if fr.digital and (p0.flags & abi.FLAG_HAS_DIGITAL_LENS):
        e |= 2
    if p0.background_mode == 3:
        e |= 16
    if p0.lens_correction_amount < 1.0:
        e |= 8
    if p0.light_refraction_coefficient != 1.0 and p0.light_refraction_coefficient > 0.0:
        e |= 4
    if np.any(np.asarray(fr.matrices)[:, 9:14] != 0.0):
        e |= 1
    return e


def _int_products_exact(n_max, n):
    if n <= 0 or n_max <= 0:
        return False
    odd = n
    while odd % 2 == 0:
        odd //= 2
    return (n_max - 1) * odd < (1 << 24)


def fused_eligible(fr):
    """the conditions under which the library serves a frame with the fused kernel at all (gfw_api.hip build_yuv_args, restated) — anything else is
    the per-plane kernel's"""
    pls = fr.planes
    p0 = pls[0]["params"]
    if pls[0]["pixel_type"] not in SAMPLE_KIND or not 1 <= len(pls) <= 4:
        return False
    bps, n0 = SAMPLE_KIND[pls[0]["pixel_type"]]
    if n0 > 1 and len(pls) != 1:
        return False
    interleaved = False
    if len(pls) >= 2:
        t1 = pls[1]["pixel_type"]
        if bps != 4 and t1 == ("UV8" if bps == 1 else "UV16"):
            if len(pls) != 2:
                return False
            interleaved = True
        elif any(pl["pixel_type"] != pls[0]["pixel_type"] for pl in pls[1:]):
            return False
    for idx, pl in enumerate(pls):
        p = pl["params"]
        if p.interpolation not in (2, 4, 8) or p.input_rotation != p0.input_rotation:
            return False
        if ((p.flags ^ p0.flags) & abi.FLAG_FIX_COLOR_RANGE) or ((p.flags & abi.FLAG_FIX_COLOR_RANGE) and p.plane_index != idx):
            return False
        if (p.flags ^ p0.flags) & abi.FLAG_FILL_WITH_BACKGROUND:
            return False
        if p.input_rotation != 0.0 and (p.lens_correction_amount < 1.0 or p.background_mode == 3):
            return False
        for st in (p.input_horizontal_stretch, p.input_vertical_stretch):
            if st > 0.001 and st != 1.0 and p.lens_correction_amount < 1.0:
                return False
        if bps in (2, 3) and ((p.stride | pl["out_size"][2]) & 1):
            return False
        if bps == 4 and ((p.stride | pl["out_size"][2]) & 3):
            return False
    if len(pls) >= 2:
        cw, ch = pls[1]["out_size"][0], pls[1]["out_size"][1]
        if p0.output_width % cw or p0.output_height % ch:
            return False
        dw, dh = p0.output_width // cw, p0.output_height // ch
        if (dw, dh) not in ((1, 1), (2, 1), (2, 2)) or (bps == 4 and (dw, dh) != (1, 1)):
            return False
        for pl in pls[1:]:
            if pl["out_size"][:2] != (cw, ch) or pl["size"][0] * dw != p0.width or pl["size"][1] * dh != p0.height:
                return False
            if pl["params"].stride != pls[1]["params"].stride or pl["out_size"][2] != pls[1]["out_size"][2]:
                return False
        if not _int_products_exact(cw, p0.output_width) or not _int_products_exact(ch, p0.output_height):
            return False
    return _int_products_exact(p0.output_width, p0.output_width) and _int_products_exact(p0.output_height, p0.output_height) and p0.width <= 65535 and p0.height <= 65535


def jit_waves(n0, matrix_count, jit_model, extras, taps, bps, dh):
    """gfw_api.hip jit_waves: the waves per SIMD a specialised build is compiled for (the kernel derives its tap rows in flight from them)"""
    if jit_model < 0 and (extras & (16 | 32)):
        return 6
    if n0 == 1 and bps <= 2 and taps == 4:
        return 6
    if n0 == 1 and bps <= 2 and taps == 8:
        return 5 if bps == 1 else 6
    return 8 if (n0 == 1 and matrix_count > 1) else 7


def run_frames(frames, mesh=None, baked=True, grid=8, votes=0, hw_ulp=0, audit=False, checksums=False):
    """One clip launch of the host-interpreted kernel over `frames` (same shape and constants, <= 16) -> [[plane outputs] per frame].
    `votes`: 0 = a wave vote answers with the lane's own predicate, 1 = as if another lane of the wave failed it (every lane takes the general route).
    `audit`: the audit instantiation (ahead-of-time form, bilinear): -> (outputs, dict of the audit words: certificates issued / wrong, queued pixels,
    largest |approximate - exact| first-pass coordinate, addresses outside the declared buffers).
    `hw_ulp`: the stand-ins for v_rcp_f32 / v_sqrt_f32 return the correctly rounded value moved by this many ulps (the hardware's are 1-ulp approximations).
    `grid`: persistent workgroups of the launch (a multiple of 8; the library launches num_cus x waves, capped at the tile count).
    `checksums`: the build that takes each frame's checksum where the pixels leave (GFW_BK_checksum, gfw_set_frame_checksums) -> (outputs, [checksum per frame]).
    baked=False: the ahead-of-time form of the same body (every clip-invariant field read from the argument block instead of a literal; one frame)."""
    fr0 = frames[0]
    p0 = fr0.planes[0]["params"]
    if audit:
        baked = False
    assert fused_eligible(fr0), "not a frame the fused kernel serves"
    bps, n0, dw, dh, il = launch_shape(fr0)
    extras = feature_bits(fr0, mesh)
    fisheye = fr0.model == abi.MODELS["opencv_fisheye"]
    # the specialised projection serves fisheye clips — also under a digital lens in a baked build, where the lens is a literal (jit_for); ahead of time
    # any feature bit sends the frame to the generic-model instantiations (gfw_kernels.hip's dispatch)
    lean = fisheye and ((extras & ~2) == 0 if baked else extras == 0)
    jit_model = 1 if lean else (-2 if extras & (16 | 32) else -1)
    stretched = any(st > 0.001 and st != 1.0 for st in (p0.input_horizontal_stretch, p0.input_vertical_stretch))      # p1_setup: exact first pass
    p1 = p1_table(p0, fr0.matrices, p0.matrix_count) if (fisheye and extras == 0 and not stretched) else None
    rform = fr0.model in RADIAL_TABLE_MODELS and extras == 0 and not stretched
    if rform:
        p1 = p1_table_radial(fr0)           # (round 6: a table over r, specialised builds only in the product; the interpreter runs either form of the body)
    fast1 = p1 is not None
    rb = 4 if fast1 else 1
    defs = {"GFW_FRAME_KIND": bps, "GFW_FRAME_TAPS": p0.interpolation, "GFW_JIT_WAVES": jit_waves(n0, p0.matrix_count, jit_model, extras, p0.interpolation, bps, dh), "GFW_JIT_MODEL": jit_model,
            "GFW_JIT_T": {1: "uint8_t", 2: "uint16_t", 3: "_Float16", 4: "float"}[bps], "GFW_JIT_N0": n0, "GFW_JIT_DW": dw, "GFW_JIT_DH": dh,
            "GFW_JIT_IL": 1 if il else 0, "GFW_JIT_RB": rb, "GFW_JIT_FAST1": 1 if fast1 else 0}
    if rform and fast1:
        defs["GFW_P1_RFORM"] = 1
    assert not checksums or (baked and not audit)
    header = _bake.bake_header(fr0, rb=rb, checksum=1 if checksums else 0)
    header, n1 = re.subn(r"#define GFW_BK_extras \(0\)", "#define GFW_BK_extras (%d)" % extras, header)
    header, n2 = re.subn(r"#define GFW_BK_digital \(0\)", "#define GFW_BK_digital (%d)" % (fr0.digital if extras & 2 else 0), header)
    assert n1 == 1 and n2 == 1
    small = len(frames) * p0.output_width * p0.output_height < 400000              # small launches: an unoptimised build compiles faster than it runs slower
    lib = C.CDLL(build(defs, header, opt="-O0" if small else "-O1", extra_flags=("-DGFW_JIT=1", "-DGFW_BAKE=%d" % (1 if baked else 0), "-DEMU_VOTES=%d" % votes, "-DEMU_HW_ULP=%d" % hw_ulp, "-DEMU_AUDIT=%d" % (1 if audit else 0))))
    lib.gfw_emu_launch.argtypes = [C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.c_void_p,
                                   C.c_float, C.c_float, C.c_float, C.c_float, C.c_float, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
    n = len(frames)
    pints, pfloats = np.zeros(24, np.int32), np.zeros(20, np.float32)        # the plane descriptors of the argument block (build_yuv_args); [16..23]: declared lengths
    for i, pl in enumerate(fr0.planes):
        q = pl["params"]
        pints[4 * i:4 * i + 4] = (q.stride, pl["out_size"][2], pl["size"][0], pl["size"][1])
        pfloats[5 * i:5 * i + 4] = [np.float32(q.background[c]) * np.float32(q.max_pixel_value) for c in range(4)]
        pfloats[5 * i + 4] = q.pixel_value_limit
        pints[16 + 2 * i], pints[16 + 2 * i + 1] = len(pl["src"]), len(pl["dst"])
    srcs, dsts, mats, keep, outs = (C.c_void_p * (4 * n))(), (C.c_void_p * (4 * n))(), (C.c_void_p * n)(), [], []
    for f, fr in enumerate(frames):
        packed = warp.pack_matrices(fr.matrices)
        if GUARD:
            packed = guarded(packed, GUARD == "end")
        keep.append(packed)
        mats[f] = packed.ctypes.data
        planes = []
        for p, pl in enumerate(fr.planes):
            if GUARD:
                src, dst = pl["src"], pl["dst"]
            else:
                src = np.concatenate([np.ascontiguousarray(pl["src"]), np.zeros(64, np.uint8)])    # slack behind the plane (see tests/test_emu_memory.py for the runs without it)
                dst = pl["dst"].copy()
            keep.append(src)
            planes.append(dst)
            srcs[4 * f + p], dsts[4 * f + p] = src.ctypes.data, dst.ctypes.data
        outs.append(planes)
    tab = p1[0] if fast1 else np.zeros((2, 2), np.float32)
    if GUARD:
        tab = guarded(tab, GUARD == "end")
    libm = C.CDLL("libm.so.6")
    libm.tanf.restype, libm.tanf.argtypes = C.c_float, [C.c_float]
    com = common_for(fr0, p0)
    if mesh is not None and len(mesh):
        mesh = np.ascontiguousarray(mesh, dtype=np.float32)
        com.mesh, com.mesh_len = mesh.ctypes.data, mesh.size
    kp = fr0.planes[0]["params"]
    sums = (C.c_ulonglong * n)()
    lib.gfw_emu_set_sums.argtypes = [C.c_void_p]
    lib.gfw_emu_set_sums(C.cast(sums, C.c_void_p) if checksums else None)
    rc = lib.gfw_emu_launch(n, srcs, dsts, mats, tab.ctypes.data, p1[1] if fast1 else 0.0, p1[2] if fast1 else 0.0, *(p1[3][:3] if fast1 else (0.0, 0.0, 0.0)),
                            C.cast(C.byref(kp), C.c_void_p), C.cast(C.byref(com), C.c_void_p), grid, pints.ctypes.data, pfloats.ctypes.data,
                            p1[4].ctypes.data if fast1 else None)
    assert rc == 0, "gfw_emu_launch -> %d" % rc
    if audit:
        words = (C.c_ulonglong * 8)()
        lib.gfw_emu_audit(words, 1)
        gap = float(np.array([int(words[4]) & 0xFFFFFFFF], dtype=np.uint32).view(np.float32)[0])
        return outs, {"certified": int(words[0]), "wrong": int(words[1]), "queued": int(words[2]), "queue_overflow": int(words[3]), "gap_px": gap,
                      "out_of_range": int(words[5]), "eps_px": float(np.array([int(words[6]) & 0xFFFFFFFF], dtype=np.uint32).view(np.float32)[0]) if fast1 else None,
                      "eps_host_px": p1[3][3] if fast1 else None}
    if checksums:
        return outs, [int(v) for v in sums]
    return outs


def run_frame(fr, mesh=None, baked=True):
    return run_frames([fr], mesh, baked)[0]


# ---- the complete per-plane operator (gfw_plane_kernel.h): every PixelType, EWA, input rotation, stretches, the colour-range and fill flags ----------

def common_for(fr, params, mesh=None):
    """gfw_api.hip fill_common restated: lens ids and the uniforms the library evaluates with the host libm (f32 arithmetic, one rounding per operation)"""
    libm = C.CDLL("libm.so.6")
    for fn in ("tanf", "cosf", "sinf", "roundf"):
        getattr(libm, fn).restype, getattr(libm, fn).argtypes = C.c_float, [C.c_float]
    f = np.float32
    com = Common(model=fr.model, digital=fr.digital, rot_cos=1.0, rot_sin=0.0, frame_w=float(params.width), frame_h=float(params.height), gopro_tt=libm.tanf(1.5533))
    if params.input_rotation != 0.0:
        rotation = f(params.input_rotation) * (f(3.14159265358979323846) / f(180.0))
        rc, rs = f(libm.cosf(rotation)), f(libm.sinf(rotation))
        s0, s1 = f(params.width), f(params.height)
        fx = rc * (s0 - f(0.0)) - rs * (s1 - f(0.0)) + f(0.0)
        fy = rs * (s0 - f(0.0)) + rc * (s1 - f(0.0)) + f(0.0)
        com.rot_cos, com.rot_sin = float(rc), float(rs)
        com.frame_w, com.frame_h = libm.roundf(abs(float(fx))), libm.roundf(abs(float(fy)))
    return com


def run_plane(fr, idx, mesh=None):
    """Plane `idx` of the frame through the host-interpreted per-plane kernel -> output bytes."""
    pl = fr.planes[idx]
    p = pl["params"]
    pix = abi.PIXEL_TYPES[pl["pixel_type"]][0]
    mesh_len = 0 if mesh is None else len(mesh)
    model = abi.MODELS["opencv_fisheye"] if (fr.model == abi.MODELS["opencv_fisheye"] and mesh_len == 0) else -1     # launch_plane_pi
    interp = p.interpolation if p.interpolation in (2, 4, 8) else 0                                                      # launch_plane_p: 10..13 = EWA
    lib = C.CDLL(build({"EMU_PIX": pix, "EMU_I": interp, "EMU_MODEL": model}, "", top="gfw_plane_kernel.h", n_asm=2, driver="emu_plane_driver.inc", extra_flags=()))
    lib.gfw_emu_launch_plane.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_longlong, C.c_int, C.c_void_p, C.c_void_p]
    com = common_for(fr, p, mesh)
    keep = None
    if mesh_len:
        keep = np.ascontiguousarray(mesh, dtype=np.float32)
        com.mesh, com.mesh_len = keep.ctypes.data, keep.size
    packed = warp.pack_matrices(fr.matrices)
    if GUARD:
        packed = guarded(packed, GUARD == "end")
        src, dst = pl["src"], pl["dst"]
    else:
        src = np.concatenate([np.ascontiguousarray(pl["src"]), np.zeros(64, np.uint8)])
        dst = pl["dst"].copy()
    rc = lib.gfw_emu_launch_plane(C.cast(C.byref(p), C.c_void_p), src.ctypes.data, dst.ctypes.data, dst.size, pl["out_size"][2], packed.ctypes.data,
                                  C.cast(C.byref(com), C.c_void_p))
    assert rc == 0, "gfw_emu_launch_plane -> %d" % rc
    return dst


def run_plane_pair(fr, idx, mesh=None):
    """Planes `idx` and `idx + 1` (U and V of a planar frame under EWA) through ONE host-interpreted launch of gfw_plane_kernel<.., DUAL> — the pairing
    gfw_api.hip run_planes makes when the two planes share every kernel parameter but plane_index and the background -> the two outputs."""
    pa, pb = fr.planes[idx], fr.planes[idx + 1]
    p = pa["params"]
    assert p.interpolation >= 10 and pa["pixel_type"] == pb["pixel_type"]
    pix = abi.PIXEL_TYPES[pa["pixel_type"]][0]
    mesh_len = 0 if mesh is None else len(mesh)
    model = abi.MODELS["opencv_fisheye"] if (fr.model == abi.MODELS["opencv_fisheye"] and mesh_len == 0) else -1
    lib = C.CDLL(build({"EMU_PIX": pix, "EMU_I": 0, "EMU_MODEL": model, "EMU_DUAL": 1}, "", top="gfw_plane_kernel.h", n_asm=2, driver="emu_plane_driver.inc", extra_flags=()))
    lib.gfw_emu_launch_plane2.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_longlong, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    com = common_for(fr, p, mesh)
    keep = None
    if mesh_len:
        keep = np.ascontiguousarray(mesh, dtype=np.float32)
        com.mesh, com.mesh_len = keep.ctypes.data, keep.size
    packed = warp.pack_matrices(fr.matrices)
    srcs = [np.concatenate([np.ascontiguousarray(pl["src"]), np.zeros(64, np.uint8)]) for pl in (pa, pb)]
    dsts = [pl["dst"].copy() for pl in (pa, pb)]
    bg2 = (C.c_float * 4)(*[float(v) for v in pb["params"].background])
    rc = lib.gfw_emu_launch_plane2(C.cast(C.byref(p), C.c_void_p), srcs[0].ctypes.data, dsts[0].ctypes.data, dsts[0].size, pa["out_size"][2], packed.ctypes.data,
                                   C.cast(C.byref(com), C.c_void_p), srcs[1].ctypes.data, dsts[1].ctypes.data, C.cast(bg2, C.c_void_p))
    assert rc == 0, "gfw_emu_launch_plane2 -> %d" % rc
    return dsts


def run_frame_per_plane(fr, mesh=None):
    return [run_plane(fr, i, mesh) for i in range(len(fr.planes))]


# ---- the coordinate kernels of gfw_kernels.hip: STMap "undist" export and the inverse point map ------------------------------------------------------

def _kernels_lib():
    lib = C.CDLL(build({}, "", top="gfw_kernels.hip", n_asm=2, driver="emu_kernels_driver.inc", extra_flags=()))
    lib.gfw_emu_stmap.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p]
    lib.gfw_emu_points.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p]
    return lib


class _Lenses:
    def __init__(self, model, digital):
        self.model, self.digital = model, digital


def stmap_undistort(params, model, digital, matrices, width, height):
    """gfw_stmap_undistort through the host-interpreted gfw_stmap_kernel -> float32 [height][width][2] (0 where the ray is rejected)."""
    coords = np.zeros((height, width, 2), dtype=np.float32)
    com = common_for(_Lenses(model, digital), params)
    packed = warp.pack_matrices(matrices)
    rc = _kernels_lib().gfw_emu_stmap(C.cast(C.byref(params), C.c_void_p), C.cast(C.byref(com), C.c_void_p), packed.ctypes.data, width, height, coords.ctypes.data)
    assert rc == 0
    return coords


def undistort_points(params, model, digital, rotations, points=None, grid=None, shifts=None, index_mode=0, mesh=None):
    """gfw_undistort_points through the host-interpreted gfw_points_kernel (argument preparation as gfw_api.hip's: cos / sin of the roll by the host libm)."""
    rot = np.ascontiguousarray(rotations, dtype=np.float32).reshape(-1, 9)
    if points is not None:
        pts = np.ascontiguousarray(points, dtype=np.float32).reshape(-1, 2)
        n, gw, pp, shape = pts.shape[0], 0, pts.ctypes.data, (pts.shape[0], 2)
    else:
        gw, gh = grid
        n, pp, shape = gw * gh, None, (gh, gw, 2)
    out = np.zeros(shape, dtype=np.float32)
    sp, packed = None, None
    if shifts is not None:
        libm = C.CDLL("libm.so.6")
        for fn in ("cosf", "sinf"):
            getattr(libm, fn).restype, getattr(libm, fn).argtypes = C.c_float, [C.c_float]
        s5 = np.ascontiguousarray(shifts, dtype=np.float32).reshape(-1, 5)
        packed = np.array([[s[0], s[1], libm.cosf(s[2]), libm.sinf(s[2]), s[3], s[4]] for s in s5], dtype=np.float32)
        sp = packed.ctypes.data
    mp, mn = None, 0
    if mesh is not None and len(mesh):
        mesh = np.ascontiguousarray(mesh, dtype=np.float64)
        mp, mn = mesh.ctypes.data, mesh.size
    com = common_for(_Lenses(model, digital), params)
    rc = _kernels_lib().gfw_emu_points(C.cast(C.byref(params), C.c_void_p), C.cast(C.byref(com), C.c_void_p), pp, n, gw, rot.ctypes.data, rot.shape[0], sp, index_mode, mp, mn, out.ctypes.data)
    assert rc == 0
    return out


# ---- the per-row matrix builder (gfw_matrices.hip) ----------------------------------------------------------------------------------------------------

def build_matrices(org, smoothed, nk, timestamps_ms, frame_readout_time_ms, rows, readout_dim, video_rotation_deg=0.0, framebuffer_inverted=False,
                   per_frame_offset_ms=0.0, offsets=None, duration_ms=1.0, suppress_rotation=0, stab=None):
    """gfw_build_matrices(_batch / _stab) through the host-interpreted kernels -> float32 [frames][rows][16] (packed rows).  Arguments as the Python
    wrappers of gyroflow_amd/warp.py take them; `stab` as a dict(offset, sensor_size, crop_area, pixel_pitch, width, height, ibis, ois)."""
    lib = C.CDLL(build({}, "", top="gfw_matrices.hip", n_asm=2, driver="emu_matrices_driver.inc", extra_flags=()))
    ts_list = list(np.atleast_1d(timestamps_ms))
    n = len(ts_list)
    arr = (abi.FrameTiming * n)()
    nkf = np.asarray(nk, dtype=np.float64).reshape(9)
    for k, ts in enumerate(ts_list):
        t = arr[k]
        t.timestamp_ms, t.per_frame_time_offset_ms, t.frame_readout_time_ms = float(ts), per_frame_offset_ms, frame_readout_time_ms
        for i in range(9):
            t.new_k[i] = nkf[i]
        t.video_rotation_deg, t.rows, t.readout_dim = video_rotation_deg, rows, readout_dim
        t.framebuffer_inverted = 1 if framebuffer_inverted else 0
        t.suppress_rotation = int(suppress_rotation)
    ot, oq = np.ascontiguousarray(org[0], dtype=np.int64), np.ascontiguousarray(org[1], dtype=np.float64)
    st, sq = np.ascontiguousarray(smoothed[0], dtype=np.int64), np.ascontiguousarray(smoothed[1], dtype=np.float64)
    fts = np.ascontiguousarray(offsets[0] if offsets else [], dtype=np.int64)
    fms = np.ascontiguousarray(offsets[1] if offsets else [], dtype=np.float64)
    out = np.zeros((n, rows, 16), dtype=np.float32)
    sd = ib = oi = None
    if stab is not None:                        # gfw_api.hip gfw_build_matrices_stab: frame_transform.rs:234-241
        inv = -1.0 if framebuffer_inverted else 1.0
        sd = np.array([stab["offset"], stab["sensor_size"][1], stab["crop_area"][1], stab["crop_area"][3],
                       stab["width"] / stab["crop_area"][2] / stab["pixel_pitch"][0], stab["height"] / stab["crop_area"][3] / stab["pixel_pitch"][1] * inv,
                       stab["height"]], dtype=np.float64)
        ib = np.ascontiguousarray(stab["ibis"], dtype=np.float64).reshape(-1, 4)
        oi = np.ascontiguousarray(stab["ois"], dtype=np.float64).reshape(-1, 4)
    lib.gfw_emu_build_matrices.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_double,
                                           C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int]
    rc = lib.gfw_emu_build_matrices(ot.ctypes.data, oq.ctypes.data, len(ot), st.ctypes.data, sq.ctypes.data, len(st),
                                    fts.ctypes.data if len(fts) else None, fms.ctypes.data if len(fts) else None, len(fts), float(duration_ms),
                                    C.cast(arr, C.c_void_p), n, rows, out.ctypes.data, rows * 16,
                                    sd.ctypes.data if sd is not None else None, ib.ctypes.data if ib is not None and len(ib) else None, len(ib) if ib is not None else 0,
                                    oi.ctypes.data if oi is not None and len(oi) else None, len(oi) if oi is not None else 0)
    assert rc == 0
    return out
