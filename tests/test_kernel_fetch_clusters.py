"""Static check of the shipped kernel cache (gyroflow_amd/jit_cache, built by tools/build_jit_cache.py with the system's hiprtc): the tap fetches of a sample must leave in
clusters — `fetch fetch fetch fetch, wait ...` — not one by one.  Round 4 found the ROCm 7.2 compiler turning unrolled fetch-and-convert loops into `fetch, wait, fetch,
wait` (20 % of a bicubic frame; profiles/r04_ab_lut_rows.txt); the source now groups the fetches explicitly, and this test reads the code objects (llvm-objdump, no device)
so that a compiler or source change which serialises them again fails at build time instead of showing up as a slower frame."""
import os
import re
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"
sys.path.insert(0, os.path.join(ROOT, "tools"))


def fetch_wait_pattern(path):
    out = subprocess.run([OBJDUMP, "-d", path], capture_output=True, text=True, check=True).stdout
    seq = []
    for line in out.splitlines():
        t = re.sub(r"^\s*[0-9a-f]+:\s*", "", line.split("//")[0]).strip()
        if t.startswith("global_load"):
            seq.append("L")
        elif t.startswith("s_waitcnt") and "vmcnt" in t:
            seq.append("w" + re.search(r"vmcnt\((\d+)\)", t).group(1))
        elif t.startswith(("s_cbranch", "s_branch")):
            seq.append("|")
        elif t.startswith("global_store"):
            seq.append("S")
    return " ".join(seq)


def shipped_entries():
    import build_jit_cache as B
    from gyroflow_amd import abi
    lib = abi.load_library()
    for label, kw in B.CONFIGS:
        _, _, name = B.key_of(lib, B.bench_frame(**kw))
        yield label, os.path.join(B.OUT, name)


@pytest.mark.skipif(not os.path.exists(OBJDUMP), reason="llvm-objdump of the ROCm toolchain not found")
def test_tap_fetches_of_the_shipped_kernels_leave_in_clusters():
    checked = 0
    for label, path in shipped_entries():
        if not os.path.exists(path):
            continue                                    # (no libhiprtc.so at build time: nothing shipped)
        pat = fetch_wait_pattern(path)
        # inside one basic block: four or more fetches each followed by a full wait
        worst = max((len(m.group(0).split()) // 2 for m in re.finditer(r"(?:L w0 ){3,}L w0", pat)), default=0)
        if "GBRAPF32LE" in label:
            # planar float frames: the three further planes' interior taps leave as one cluster of six on the wave-uniform path (gfw_frame.hip, "the EXR route");
            # the fallback of a wave with a border sample (sample_store_shared_refs, plane after plane) is the one place where a fetch is waited for by itself —
            # border waves only, none in the C4 crop
            assert worst <= 6 and re.search(r"(?:L ){6}w5", pat), "%s: %d fetches in a row each waited for by itself / no cluster of six\n%s" % (label, worst, pat)
        else:
            assert worst < 4, "%s: %d fetches in a row each waited for by itself\n%s" % (label, worst, pat)
        if "bicubic" in label or "Lanczos" in label:
            assert re.search(r"(?:L ){4,}w", pat), "%s: no cluster of four tap-row fetches\n%s" % (label, pat)
        checked += 1
    if checked == 0:
        pytest.skip("no shipped kernel cache in this tree")
