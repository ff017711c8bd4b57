"""The 32-phase tap table: generator output == committed tables == the reference's literal table (when mounted)."""
import os
import re
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import gen_coeffs  # noqa: E402

REF = "/root/reference/src/core/stabilization/cpu_undistort.rs"


def parse_inc(path):
    body = open(path).read()
    body = body[body.index("{") + 1: body.rindex("}")]
    return np.array([float(t.rstrip("f")) for t in body.replace("\n", " ").split(",") if t.strip()], dtype=np.float32)


def test_committed_tables_match_generator():
    want = np.array(gen_coeffs.table(), dtype=np.float32)
    assert want.size == 448
    for rel in ("oracle/gfw_coeffs.inc", "gyroflow_amd/csrc/gfw_coeffs.inc"):
        got = parse_inc(os.path.join(ROOT, rel))
        assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), rel


def test_bilinear_rows_are_exact_k_over_32():
    t = np.array(gen_coeffs.table(), dtype=np.float32)[:64].reshape(32, 2)
    k = np.arange(32, dtype=np.float32)
    assert np.array_equal(t[:, 1], k * np.float32(0.03125)) and np.array_equal(t[:, 0], np.float32(1.0) - k * np.float32(0.03125))


def test_table_equals_reference_literals_when_mounted():
    if not os.path.exists(REF):
        import pytest
        pytest.skip("reference tree not mounted on this box")
    src = open(REF).read()
    m = re.search(r"pub const COEFFS.*?= \[(.*?)\];", src, re.S)
    body = re.sub(r"//.*", "", m.group(1))
    ref = np.array([float(t) for t in body.replace("\n", " ").split(",") if t.strip()], dtype=np.float32)[:448]
    ours = np.array(gen_coeffs.table(), dtype=np.float32)
    assert np.array_equal(ref.view(np.uint32), ours.view(np.uint32))
