"""The oracle is test infrastructure: nothing under gyroflow_amd/ (Python or C++/HIP) may import, link or mention it, the
shared library must not depend on it, and the product must fail loudly without its HIP extension."""
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "gyroflow_amd")


def product_files():
    for d, _, files in os.walk(PKG):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".hpp", ".cpp", ".inc")):
                yield os.path.join(d, f)


def test_product_sources_never_reference_the_oracle():
    pat = re.compile(r"gfw_oracle|_oracle\b|oracle/|import\s+_oracle|libgfw_oracle")
    hits = []
    for path in product_files():
        for n, line in enumerate(open(path, errors="replace"), 1):
            if pat.search(line):
                hits.append("%s:%d: %s" % (os.path.relpath(path, ROOT), n, line.strip()))
    assert not hits, "\n".join(hits)


def test_product_sources_never_reference_the_host_interpreter_or_the_reference_build():
    """tests/_emu.py (the kernels' source interpreted on the host) and oracle/_ref (the reference's own kernel, host build) are test infrastructure too:
    no product file may name them, and bench.py may touch neither (its only CPU legs are the oracle's parity check and cpu_baseline)."""
    pat = re.compile(r"\b_emu\b|tests/emu|emu_prelude|gfw_emu_|EMU_VOTES|EMU_HW_ULP|oracle/_ref|gfw_ref_cl|ref_cl_host")
    hits = []
    for path in list(product_files()) + [os.path.join(ROOT, "bench.py"), os.path.join(ROOT, "include", "gfwarp.h"), os.path.join(ROOT, "include", "gfwarp.hpp")]:
        for n, line in enumerate(open(path, errors="replace"), 1):
            if pat.search(line):
                hits.append("%s:%d: %s" % (os.path.relpath(path, ROOT), n, line.strip()))
    assert not hits, "\n".join(hits)


def test_shared_library_does_not_link_the_oracle():
    lib = os.path.join(PKG, "libgfwarp.so")
    if not os.path.exists(lib):
        pytest.skip("libgfwarp.so not built")
    deps = subprocess.check_output(["readelf", "-d", lib]).decode()
    assert "oracle" not in deps
    syms = subprocess.check_output(["nm", "-D", "--defined-only", lib]).decode()
    assert "gfw_oracle" not in syms


def test_missing_extension_is_a_loud_failure(tmp_path):
    from gyroflow_amd import abi
    with pytest.raises((OSError, RuntimeError)) as e:
        abi.load_library(str(tmp_path / "libgfwarp.so"))
    assert "no CPU fallback" in str(e.value) or "libgfwarp" in str(e.value)
