"""Test-side restatement of gfw_api.hip's bake header (GFW_BK_* literals of a frame's clip-invariant arguments) for host-side build
checks of the run-time specialisation path; the library generates the real one from the arguments it validated."""
import struct


def _f(v):
    return "__builtin_bit_cast(float, 0x%08xu)" % struct.unpack("<I", struct.pack("<f", float(v)))[0]


def rotated_frame_size(p0):
    """cpu_undistort.rs:484-490 in f32 with the host libm (as gfw_api.hip fill_common evaluates it): the source frame turned by input_rotation"""
    import ctypes as C
    import numpy as np
    if p0.input_rotation == 0.0:
        return float(p0.width), float(p0.height)
    libm = C.CDLL("libm.so.6")
    for fn in ("cosf", "sinf", "roundf"):
        getattr(libm, fn).restype, getattr(libm, fn).argtypes = C.c_float, [C.c_float]
    f = np.float32
    rotation = f(p0.input_rotation) * (f(3.14159265358979323846) / f(180.0))
    rc, rs = f(libm.cosf(rotation)), f(libm.sinf(rotation))
    s0, s1 = f(p0.width), f(p0.height)
    fx = rc * (s0 - f(0.0)) - rs * (s1 - f(0.0)) + f(0.0)
    fy = rs * (s0 - f(0.0)) + rc * (s1 - f(0.0)) + f(0.0)
    return libm.roundf(abs(float(fx))), libm.roundf(abs(float(fy)))


def bake_header(frame, rb=4, checksum=0):
    pls = frame.planes
    p0 = pls[0]["params"]
    w, h = p0.width, p0.height
    ow, oh = p0.output_width, p0.output_height
    n = len(pls)
    if n >= 2:
        dw, dh = ow // pls[1]["out_size"][0], oh // pls[1]["out_size"][1]
    else:
        dw = dh = 1
    cw, ch = (ow + dw - 1) // dw, (oh + dh - 1) // dh
    d = {"nplanes": n, "width": w, "height": h, "out_w": ow, "out_h": oh, "cw": cw, "ch": ch, "tiles_x": (cw + 63) // 64,
         "tiles_y": (ch + 4 * rb - 1) // (4 * rb), "matrix_count": p0.matrix_count, "hrs": 1 if p0.flags & 16 else 0, "model": frame.model,
         "k_all_zero": 1 if all(p0.k[i] == 0.0 for i in range(4)) else 0, "background_mode": p0.background_mode, "extras": 0, "ablate": 0, "digital": 0,
         "fill_bg": 1 if p0.flags & 4 else 0, "rot_on": 1 if p0.input_rotation != 0.0 else 0, "fix_range": 1 if p0.flags & 1 else 0, "checksum": checksum,
         "hstretch_div": 1 if (p0.input_horizontal_stretch > 0.001 and p0.input_horizontal_stretch != 1.0) else 0,
         "vstretch_div": 1 if (p0.input_vertical_stretch > 0.001 and p0.input_vertical_stretch != 1.0) else 0}
    out = ["#define GFW_BK_%s (%d)" % kv for kv in d.items()]
    out.append("#define GFW_BK_audit ((unsigned long long *)nullptr)")
    fl = {"f_0": p0.f[0], "f_1": p0.f[1], "c_0": p0.c[0], "c_1": p0.c[1], "t2_0": p0.translation2d[0], "t2_1": p0.translation2d[1],
          "hstretch": p0.input_horizontal_stretch, "vstretch": p0.input_vertical_stretch, "k_0": p0.k[0], "k_1": p0.k[1], "k_2": p0.k[2], "k_3": p0.k[3], "r_limit_sq": struct.unpack("<f", struct.pack("<f", p0.r_limit))[0] ** 2}
    hrs = d["hrs"]
    fl["p1_f"], fl["p1_c"] = (p0.f[0], p0.c[0]) if hrs else (p0.f[1], p0.c[1])
    import numpy as np
    one = np.float32(1.0)
    cpl = pls[1] if n >= 2 else pls[0]
    fw, fh = rotated_frame_size(p0)
    for name, mul, den in (("map_lx", pls[0]["size"][0], fw), ("map_ly", pls[0]["size"][1], fh), ("map_cx", cpl["size"][0], fw), ("map_cy", cpl["size"][1], fh)):
        fl[name + "_mul"], fl[name + "_den"], fl[name + "_rcp"] = float(mul), float(den), float(one / np.float32(den))
    for i in range(4):
        if i < n:
            p, pl = pls[i]["params"], pls[i]
            vals = {"src_stride": p.stride, "dst_stride": pl["out_size"][2], "w": pl["size"][0], "h": pl["size"][1],
                    "fix": (1 if p.plane_index == 0 else 2) if p.flags & 1 else 0, "ox32": 0, "oy32": 0}
            bg = [float(np.float32(p.background[c]) * np.float32(p.max_pixel_value)) for c in range(4)]
            lim = p.pixel_value_limit
        else:
            vals, bg, lim = {"src_stride": 0, "dst_stride": 0, "w": 0, "h": 0, "fix": 0, "ox32": 0, "oy32": 0}, [0.0] * 4, 0.0
        out += ["#define GFW_BK_pl%d_%s (%d)" % (i, k, v) for k, v in vals.items()]
        for c in range(4):
            fl["pl%d_bg_%d" % (i, c)] = bg[c]
        fl["pl%d_limit" % i] = lim
        fl["pl%d_org_x" % i] = fl["pl%d_org_y" % i] = 0.0          # (the interpreter's frames have no buffer rects; the GPU tier tests them)
    out += ["#define GFW_BK_%s %s" % (k, _f(v)) for k, v in fl.items()]
    return "\n".join(out) + "\n"
