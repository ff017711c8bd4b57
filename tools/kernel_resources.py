#!/usr/bin/env python3
"""Registers / LDS / scratch of every kernel in a HIP shared library, read from the code objects' metadata notes.

The gfx950 code objects sit in the library's .hip_fatbin section; each is an ELF64 (EM_AMDGPU) whose NT_AMDGPU_METADATA note
is a msgpack map (`amdhsa.kernels`).  No ROCm tool is needed.  usage: kernel_resources.py [lib.so] [name-filter]"""
import struct
import sys

import msgpack


def code_objects(blob):
    pos = 0
    while True:
        pos = blob.find(b"\x7fELF\x02\x01", pos)
        if pos < 0:
            return
        e_machine = struct.unpack_from("<H", blob, pos + 18)[0]
        if e_machine == 224:                                       # EM_AMDGPU
            e_shoff, = struct.unpack_from("<Q", blob, pos + 40)
            e_shentsize, e_shnum = struct.unpack_from("<HH", blob, pos + 58)
            end = pos + e_shoff + e_shentsize * e_shnum
            yield blob[pos:end]
            pos = end
        else:
            pos += 4


def kernels_of(elf):
    e_shoff, = struct.unpack_from("<Q", elf, 40)
    e_shentsize, e_shnum = struct.unpack_from("<HH", elf, 58)
    for i in range(e_shnum):
        sh = e_shoff + i * e_shentsize
        sh_type, = struct.unpack_from("<I", elf, sh + 4)
        off, size = struct.unpack_from("<QQ", elf, sh + 24)
        if sh_type != 7:                                           # SHT_NOTE
            continue
        p = off
        while p + 12 <= off + size:
            namesz, descsz, ntype = struct.unpack_from("<III", elf, p)
            name = elf[p + 12:p + 12 + namesz].rstrip(b"\0")
            d0 = p + 12 + ((namesz + 3) & ~3)
            if name == b"AMDGPU" and ntype == 32:
                md = msgpack.unpackb(elf[d0:d0 + descsz], raw=False, strict_map_key=False)
                for k in md.get("amdhsa.kernels", []):
                    yield k
            p = d0 + ((descsz + 3) & ~3)


def waves_per_simd(vgpr):
    alloc = max(8, (vgpr + 7) // 8 * 8)
    return min(8, 512 // alloc)


def workgroups_per_cu(k, threads=256):
    """256-thread workgroups a CU admits: VGPRs, SGPRs (MI355X_MICROARCH.md: floor(800 / (ceil(sgpr/16)*16 + 16))), LDS."""
    by_v = waves_per_simd(k[".vgpr_count"])
    by_s = min(8, 800 // (((k[".sgpr_count"] + 15) // 16) * 16 + 16))
    lds = k[".group_segment_fixed_size"]
    by_l = 8 if lds == 0 else min(8, 163840 // lds)
    return min(by_v, by_s, by_l)


def report(path):
    blob = open(path, "rb").read()
    rows = []
    for elf in code_objects(blob):
        for k in kernels_of(elf):
            rows.append(k)
    return rows


if __name__ == "__main__":
    path = sys.argv[1] if len(sys.argv) > 1 else "gyroflow_amd/libgfwarp.so"
    flt = sys.argv[2] if len(sys.argv) > 2 else ""
    print("%-6s %-6s %-8s %-8s %-7s %s" % ("vgpr", "sgpr", "lds", "scratch", "wg/CU", "kernel"))
    for k in sorted(report(path), key=lambda k: k[".name"]):
        if flt in k[".name"]:
            print("%-6d %-6d %-8d %-8d %-7d %s" % (k[".vgpr_count"], k[".sgpr_count"], k[".group_segment_fixed_size"],
                                                   k[".private_segment_fixed_size"], workgroups_per_cu(k), k[".name"]))
