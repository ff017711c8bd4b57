"""Device arithmetic vs the host's libm / IEEE operations (run on the GPU box)."""
import ctypes as C

import numpy as np
import pytest

from gyroflow_amd import abi
import _oracle as O

pytestmark = pytest.mark.gpu


def dev(op, a, b=None):
    lib = abi.load_library()
    a = np.ascontiguousarray(a, dtype=np.float32)
    out = np.empty_like(a)
    bp = None
    if b is not None:
        b = np.ascontiguousarray(b, dtype=np.float32)
        bp = b.ctypes.data
    rc = lib.gfw_debug_math(op, a.ctypes.data, bp, out.ctypes.data, a.size)
    assert rc == 0, lib.gfw_last_error()
    return out


def libm(fn, a):
    a = np.ascontiguousarray(a, dtype=np.float32)
    out = np.empty_like(a)
    O.lib().gfw_oracle_libm(fn, a.ctypes.data, out.ctypes.data, a.size)
    return out


def same_bits(x, y):
    nan = np.isnan(x) & np.isnan(y)
    return np.all((x.view(np.uint32) == y.view(np.uint32)) | nan)


def sample_floats(n, seed, lo_exp=-30, hi_exp=30):
    rng = np.random.default_rng(seed)
    bits = rng.integers(0, 1 << 32, size=n, dtype=np.uint64).astype(np.uint32)
    x = bits.view(np.float32)
    e = rng.integers(lo_exp, hi_exp, size=n)
    m = rng.random(n, dtype=np.float32) + np.float32(1.0)
    y = (m * np.exp2(e).astype(np.float32)) * np.where(rng.random(n) < 0.5, -1, 1).astype(np.float32)
    return np.concatenate([x[: n // 2], y[n // 2:], np.array([0.0, -0.0, np.inf, -np.inf, np.nan, 1.0, 0.4375, 0.6875, 1.1875, 2.4375, 3.3554432e7], dtype=np.float32)])


def test_device_atanf_equals_libm():
    x = sample_floats(1 << 22, 1)
    assert same_bits(dev(0, x), libm(0, x))
    # dense sweep of the range lens projections use
    x = np.linspace(0, 4, 1 << 22, dtype=np.float32)
    assert same_bits(dev(0, x), libm(0, x))
    assert same_bits(dev(2, x), libm(0, x))


def test_device_tanf_equals_libm():
    x = sample_floats(1 << 22, 2)
    assert same_bits(dev(1, x), libm(1, x))
    x = np.linspace(-3.2, 3.2, 1 << 22, dtype=np.float32)
    assert same_bits(dev(1, x), libm(1, x))
    x = np.linspace(100.0, 1e6, 1 << 20, dtype=np.float32)         # reduce_large path
    assert same_bits(dev(1, x), libm(1, x))


def test_device_division_and_sqrt_are_ieee():
    a = sample_floats(1 << 21, 3)
    b = sample_floats(1 << 21, 4)
    with np.errstate(all="ignore"):
        assert same_bits(dev(4, a, b), (a / b).astype(np.float32))
        assert same_bits(dev(6, np.abs(a)), np.sqrt(np.abs(a)))


def test_casts_and_round_follow_rust_semantics():
    x = np.array([0.0, -0.0, 0.5, -0.5, 1.5, 2.5, -2.5, 0.49999997, 8388609.0, 65535.9, 65536.0, -1.0, 255.5, 256.0, 3e9, -3e9,
                  np.inf, -np.inf, np.nan, 1e-40], dtype=np.float32)
    r = dev(9, x)
    exp_round = np.array([0, -0.0, 1, -1, 2, 3, -3, 0, 8388609, 65536, 65536, -1, 256, 256, 3e9, -3e9, np.inf, -np.inf, np.nan, 0], dtype=np.float32)
    assert same_bits(r, exp_round)
    i32 = dev(7, x)
    assert list(i32[:16]) == [0, 0, 0, 0, 1, 2, -2, 0, 8388609, 65535, 65536, -1, 255, 256, 2147483648.0, -2147483648.0]
    assert list(i32[16:19]) == [2147483648.0, -2147483648.0, 0.0]          # inf saturates, NaN -> 0
    u16 = dev(8, x)
    assert list(u16[:16]) == [0, 0, 0, 0, 1, 2, 0, 0, 65535, 65535, 65535, 0, 255, 256, 65535, 0]
    assert list(u16[16:19]) == [65535, 0, 0]
    u8 = dev(10, x)
    assert list(u8[:16]) == [0, 0, 0, 0, 1, 2, 0, 0, 255, 255, 255, 0, 255, 255, 255, 0]


def test_lean_primitives_equal_generic_on_a_billion_operands():
    lib = abi.load_library()
    assert lib.gfw_debug_selftest(0, 1 << 30, 12345) == 0, "lean divide"
    assert lib.gfw_debug_selftest(1, 1 << 30, 67890) == 0, "lean sqrt"
    assert lib.gfw_debug_selftest(2, 0, 0) == 0, "atanf_pos vs atanf over all non-negative floats"
    # the one-correction divide over significand pairs: 16384 denominators x every 64th numerator here (1.5e9 quotients);
    # tools/prove_div.py runs all 7.04e13 pairs (profiles/r01_div_one_correction_proof.txt)
    assert lib.gfw_debug_selftest(3, 1 << 14, (6 << 32) | 0x2a5f13) == 0, "refined reciprocal + one correction"


def test_device_sinf_cosf_equal_libm():
    for seed, lo, hi in ((11, -30, 8), (12, -4, 40)):
        x = sample_floats(1 << 21, seed, lo, hi)
        assert same_bits(dev(11, x), libm(2, x))
        assert same_bits(dev(12, x), libm(3, x))
    x = np.linspace(-7.0, 7.0, 1 << 22, dtype=np.float32)
    assert same_bits(dev(11, x), libm(2, x))
    assert same_bits(dev(12, x), libm(3, x))
    x = np.linspace(100.0, 3e7, 1 << 20, dtype=np.float32)          # reduce_large path
    assert same_bits(dev(11, x), libm(2, x))
    assert same_bits(dev(12, x), libm(3, x))
