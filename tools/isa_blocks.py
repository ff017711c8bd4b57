#!/usr/bin/env python3
"""Where the instructions of the specialised kernel sit: compiles the text libgfwarp.so embeds for hiprtc (tools/gen_jit_source.py) offline for gfx950 with
line tables — same flags as gfw_jit.hip, same bake header (default: tools/bake_c2.h, the C2 clip) — and prints, per loop nest and per basic block, the VALU /
memory instruction counts with the source line that dominates the block.  No device needed (hipcc cross-compiles).

usage: tools/isa_blocks.py [--defs "GFW_FRAME_TAPS=8;..."] [--header bake.h] [--loop BB0_95]
  without --loop: the loop nests of gfw_jit_kernel with their static VALU counts; with it: the blocks of that loop in layout order.
The hot path of the phase-3 loop (C2, round 3): matrix-vector 19, lean divide + r^2 + sqrt 34, table atanf 41 (20 when the whole wave is below 0.4375),
theta polynomial + divide 21, scale 3, source_rect map + bins 30, interior taps 12 per luma pixel; chroma site + stores 59; first pass 20 per pixel."""
import argparse
import collections
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import gen_jit_source as G  # noqa: E402

C2_DEFS = "GFW_FRAME_KIND=2;GFW_FRAME_TAPS=2;GFW_JIT_WAVES=8;GFW_JIT_MODEL=1;GFW_JIT_T=uint16_t;GFW_JIT_N0=1;GFW_JIT_DW=2;GFW_JIT_DH=1;GFW_JIT_IL=0;GFW_JIT_RB=4;GFW_JIT_FAST1=1"


def compile_isa(defs, header, workdir):
    out = []
    G.expand(os.path.join(G.CSRC, "gfw_frame.hip"), set(), out)
    hip, asm = os.path.join(workdir, "k.hip"), os.path.join(workdir, "k.s")
    open(hip, "w").write("#include <hip/hip_runtime.h>\n#include <stdint.h>\n" + open(header).read() + "\n" + "".join(out))
    d = dict(kv.split("=", 1) for kv in C2_DEFS.split(";"))
    flags = [kv for kv in defs.split(";") if kv.startswith("-")]              # compiler options travel like GFW_JIT_DEFS entries that start with '-'
    d.update(kv.split("=", 1) for kv in defs.split(";") if kv and not kv.startswith("-"))
    cmd = ["/opt/rocm/bin/hipcc", "-x", "hip", "--offload-arch=gfx950", "--cuda-device-only", "-S", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-slp-vectorize", "-Wno-pass-failed",
           "-Wno-cuda-compat", "-gline-tables-only", "-DGFW_JIT=1", "-DGFW_BAKE=1"] + flags + ["-D%s=%s" % kv for kv in d.items()] + [hip, "-o", asm]
    subprocess.run(cmd, check=True, capture_output=True)
    return open(asm).read().splitlines(), open(hip).read().splitlines()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--defs", default="")
    ap.add_argument("--header", default=os.path.join(ROOT, "tools", "bake_c2.h"))
    ap.add_argument("--loop", default="")
    a = ap.parse_args()
    with tempfile.TemporaryDirectory() as td:
        lines, src = compile_isa(a.defs, a.header, td)
    start = next(i for i, l in enumerate(lines) if l.startswith("gfw_jit_kernel:"))
    end = next(i for i in range(start, len(lines)) if lines[i].strip().startswith(".Lfunc_end"))
    blocks, cur, loc = [], None, None
    for l in lines[start:end]:
        m = re.match(r"^\.LBB0_(\d+):\s*(?:;\s*(.*))?$", l)
        if m:
            info = m.group(2) or ""
            mm = re.search(r"in Loop: Header=(BB0_\d+) Depth=(\d+)", info)
            loop = ("BB0_" + m.group(1), int(re.search(r"Depth=(\d+)", info).group(1))) if "Loop Header" in info else ((mm.group(1), int(mm.group(2))) if mm else ("-", 0))
            cur = {"name": "BB0_" + m.group(1), "loop": loop, "valu": 0, "mem": 0, "lines": collections.Counter(), "term": ""}
            blocks.append(cur)
            continue
        mm = re.match(r"\s*\.loc\s+\d+\s+(\d+)", l)
        if mm:
            loc = int(mm.group(1))
            continue
        t = l.strip().split(";")[0].strip()
        if cur is None or not t or t.startswith("."):
            continue
        if t.startswith("v_"):
            cur["valu"] += 1
            cur["lines"][loc] += 1
        if t.startswith(("global_", "ds_", "buffer_")):
            cur["mem"] += 1
        if t.startswith(("s_cbranch", "s_branch")):
            cur["term"] += t.split()[0].replace("s_cbranch_", "") + "->" + t.split()[-1].replace(".LBB0_", "") + " "
    if not a.loop:
        tot = collections.Counter()
        for b in blocks:
            tot[b["loop"]] += b["valu"]
        for k, v in sorted(tot.items(), key=lambda kv: -kv[1]):
            print("loop %-8s depth %d: %4d VALU in %d blocks" % (k[0], k[1], v, sum(1 for b in blocks if b["loop"] == k)))
        return
    for b in blocks:
        if b["loop"][0] != a.loop:
            continue
        top = b["lines"].most_common(1)
        ln = top[0][0] if top else None
        print("%-8s valu %3d mem %2d  %-34s %s" % (b["name"], b["valu"], b["mem"], b["term"][:34], src[ln - 1].strip()[:100] if ln else ""))


if __name__ == "__main__":
    main()
