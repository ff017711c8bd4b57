#!/usr/bin/env python3
"""gfx950: a VALU instruction that reads a VGPR written by a transcendental (v_rcp / v_rsq / v_sqrt / v_exp / v_log / v_sin / v_cos) needs a wait state in between.
The compiler's hazard recogniser inserts it for its own instructions but cannot see inside an `asm` statement — round 6's r-form first pass read a stale register
through `v_sqrt_f32` followed at once by min_limit's inline `v_min_f32` (profiles/r06_gopro_first_pass.txt).  This scans the device code of a library / of code
objects for the pattern — a transcendental whose destination is a source of the very next instruction — so that no other asm statement carries it.
usage: tools/scan_trans_hazard.py [libgfwarp.so | file.co ...]      exit status 1 when the pattern is found"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import kernel_resources as KR  # noqa: E402

OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"
TRANS = re.compile(r"^\s*v_(rcp|rsq|sqrt|exp|log|sin|cos)_(f32|f16|legacy_f32|iflag_f32)\w*\s+(v\d+|v\[\d+:\d+\])")


def scan(elf_bytes, label):
    with tempfile.NamedTemporaryFile(suffix=".co") as f:
        f.write(elf_bytes); f.flush()
        txt = subprocess.run([OBJDUMP, "-d", f.name], capture_output=True, text=True).stdout
    found, prev, kernel = [], None, "?"
    for line in txt.splitlines():
        m = re.match(r"^[0-9a-f]+ <(.+)>:", line)
        if m:
            kernel, prev = m.group(1), None
            continue
        ins = line.split("//")[0].strip()
        if not ins or ins.endswith(":"):
            continue
        if prev:
            dst = prev.group(3)
            ops = ins.split(None, 1)
            if ops[0].startswith("v_") and len(ops) > 1:
                srcs = ops[1].split(",")[1:] if not ops[0].startswith(("v_cmp", "v_cmpx")) else ops[1].split(",")
                if any(re.search(r"\b%s\b" % re.escape(dst), s) for s in srcs) and not ops[0].startswith(("v_rcp", "v_rsq", "v_sqrt", "v_exp", "v_log", "v_sin", "v_cos")):
                    found.append("%s: %s  ->  %s" % (kernel[:70], prev.group(0).strip(), ins))
        prev = TRANS.match(ins)
    return found


def main():
    paths = sys.argv[1:] or [os.path.join(ROOT, "gyroflow_amd", "libgfwarp.so")] + sorted(
        os.path.join(ROOT, "gyroflow_amd", "jit_cache", f) for f in os.listdir(os.path.join(ROOT, "gyroflow_amd", "jit_cache")) if f.endswith(".co"))
    import concurrent.futures
    bad = 0
    jobs = []
    for p in paths:
        blob = open(p, "rb").read()
        jobs += [(p, i, co) for i, co in enumerate(KR.code_objects(blob))]
    with concurrent.futures.ThreadPoolExecutor(max_workers=min(8, os.cpu_count() or 1)) as ex:
        for (p, i, _), found in zip(jobs, ex.map(lambda j: scan(j[2], j[0]), jobs)):
            for f in found:
                bad += 1
                if bad <= 40:
                    print("%s [code object %d] %s" % (os.path.basename(p), i, f))
    print("%d code object(s) of %d file(s) scanned" % (len(jobs), len(paths)))
    print("transcendental result read by the next instruction: %d place(s)" % bad)
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
