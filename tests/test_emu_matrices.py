"""The per-row matrix builder's kernels (gfw_matrices.hip: quaternion lookup + nalgebra-order slerp + closed-form inverse in f64, IBIS/OIS Catmull-Rom terms,
suppress_rotation, sync offsets) host-interpreted (tests/_emu.py) against the independent f64 host statement of frame_transform.rs:221-308
(tests/_hoststmt.py) — the CPU-tier twin of tests/test_gpu_matrix_builder.py, same inputs, same bar (<= 2 ULP of f32 on the matrix entries, <= 1 ULP on the
stabiliser terms, cos / sin slots exactly the host libm's of the f32 angle) — and, fed to the interpreted warp kernel, bit-exact against the oracle fed the same rows."""
import numpy as np
import pytest

from gyroflow_amd import synthetic as S
import _emu
import _hoststmt as HS
import _oracle as O
from test_gpu_matrix_builder import ulps, _stab

W, H = 640, 360


def tracks():
    return S.sampled_track(11, 0.0, 2000.0, 1000.0), S.sampled_track(12, 0.0, 2000.0, 200.0, scale=0.25)


@pytest.mark.parametrize("readout_ms,inverted,rot", [(16.0, False, 0.0), (-12.0, False, 0.0), (8.0, True, 0.0), (16.0, False, 90.0), (0.0, False, 0.0)])
def test_rows_match_the_host_f64_statement(readout_ms, inverted, rot):
    fr = S.SyntheticFrame("YUV422P16LE", W, H, seed=3, pixels=False)
    org, sm = tracks()
    nk = S.new_k(fr.lens, 1.0, W, H)
    rows = H if abs(readout_ms) > 0 else 1
    host = HS.row_matrices_from_tracks(org, sm, nk, 1000.3, readout_ms, rows, H, rot, inverted, 0.2)
    dev = _emu.build_matrices(org, sm, nk, 1000.3, readout_ms, rows, H, rot, inverted, 0.2)[0]
    assert np.all(dev[:, 9:14] == 0) and np.all(dev[:, 14] == 1) and np.all(dev[:, 15] == 0)
    scale = np.abs(host[:, :9]).max(axis=1, keepdims=True) * 1e-4
    assert ulps(dev[:, :9], host[:, :9], scale).max() <= 2.0


@pytest.mark.parametrize("inverted", [False, True])
def test_ibis_ois_spline_terms_per_row(inverted):
    fr = S.SyntheticFrame("YUV422P16LE", W, H, seed=3, pixels=False)
    org, sm = tracks()
    nk = S.new_k(fr.lens, 1.0, W, H)
    stab = _stab(W, H)
    host = HS.row_matrices_from_tracks(org, sm, nk, 1000.3, 16.0, H, H, framebuffer_inverted=inverted, stab=stab)
    dev = _emu.build_matrices(org, sm, nk, 1000.3, 16.0, H, H, framebuffer_inverted=inverted, stab=stab)[0]
    assert np.abs(host[:, 9:14]).max() > 0.5
    assert ulps(dev[:, 9:14], host[:, 9:14], np.full((H, 1), 1e-6)).max() <= 1.0
    lib = O.lib()
    ang = np.ascontiguousarray(-dev[:, 11], dtype=np.float32)
    c, sn = np.empty_like(ang), np.empty_like(ang)
    lib.gfw_oracle_libm(3, ang.ctypes.data, c.ctypes.data, ang.size)
    lib.gfw_oracle_libm(2, ang.ctypes.data, sn.ctypes.data, ang.size)
    assert np.array_equal(dev[:, 14].view(np.uint32), c.view(np.uint32)) and np.array_equal(dev[:, 15].view(np.uint32), sn.view(np.uint32))
    scale = np.abs(host[:, :9]).max(axis=1, keepdims=True) * 1e-4
    assert ulps(dev[:, :9], host[:, :9], scale).max() <= 2.0


@pytest.mark.parametrize("mode", [1, 2])
def test_suppress_rotation(mode):
    fr = S.SyntheticFrame("YUV422P16LE", W, H, seed=3, pixels=False)
    org, sm = tracks()
    nk = S.new_k(fr.lens, 1.0, W, H)
    stab = _stab(W, H)
    host = HS.row_matrices_from_tracks(org, sm, nk, 1000.3, 16.0, H, H, suppress_rotation=mode, stab=stab)
    dev = _emu.build_matrices(org, sm, nk, 1000.3, 16.0, H, H, suppress_rotation=mode, stab=stab)[0]
    assert np.all(dev[:, :9] == dev[0, :9])
    scale = np.abs(host[:, :9]).max(axis=1, keepdims=True) * 1e-4
    assert ulps(dev[:, :9], host[:, :9], scale).max() <= 2.0
    if mode == 2:
        assert np.all(dev[:, 9:14] == 0) and np.all(dev[:, 14] == 1) and np.all(dev[:, 15] == 0)
    else:
        assert ulps(dev[:, 9:14], host[:, 9:14], np.full((H, 1), 1e-6)).max() <= 1.0


def test_sync_offsets_and_a_batch_of_frames():
    fr = S.SyntheticFrame("YUV422P16LE", W, H, seed=3, pixels=False)
    org, sm = tracks()
    nk = S.new_k(fr.lens, 1.0, W, H)
    offsets = (np.array([0, 700000, 1500000], dtype=np.int64), np.array([3.5, -2.25, 6.0]))
    stamps = [1000.3 + 33.3 * j for j in range(4)]
    dev = _emu.build_matrices(org, sm, nk, stamps, 16.0, H, H, offsets=offsets, duration_ms=2000.0)
    for j, ts in enumerate(stamps):
        host = HS.row_matrices_from_tracks(org, sm, nk, ts, 16.0, H, H, offsets=offsets, duration_ms=2000.0)
        scale = np.abs(host[:, :9]).max(axis=1, keepdims=True) * 1e-4
        assert ulps(dev[j][:, :9], host[:, :9], scale).max() <= 2.0, j


def test_warp_with_interpreter_built_rows_is_bit_exact_against_the_oracle_fed_the_same_rows():
    fr = S.SyntheticFrame("YUV422P16LE", W, H, seed=9)
    org = S.sampled_track(21, 0.0, 2000.0, 1000.0)
    sm = S.sampled_track(22, 0.0, 2000.0, 200.0, scale=0.25)
    nk = S.new_k(fr.lens, 1.0, W, H)
    rows = _emu.build_matrices(org, sm, nk, 987.6, 16.0, H, H, stab=_stab(W, H))[0]
    fr.matrices = np.ascontiguousarray(rows[:, :14])
    for a, b in zip(O.run_frame(fr), _emu.run_frame(fr)):
        assert np.array_equal(a, b)
