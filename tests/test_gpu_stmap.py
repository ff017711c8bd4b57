"""STMap 'undist' coordinate export (gfw_stmap_undistort) vs the oracle restatement of src/core/stmap.rs:87-109."""
import numpy as np
import pytest

from gyroflow_amd import abi, synthetic as S, warp
import _oracle as O

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("model,k", [("opencv_fisheye", [0.045, 0.02, -0.02, 0.006]), ("opencv_standard", [0.1, -0.05, 0.001, 0.002, 0.01, 0.0, 0.0, 0.0]),
                                     ("poly5", [0.08, -0.02]), ("sony", [1.0, 0.01, -0.05, 0.02, 0.0, 0.0]), ("gopro", [0.0, 1.0, 0.01, -0.12, 0.02, 0.01, -0.004])])
@pytest.mark.parametrize("hrs", [False, True])
def test_stmap_matches_oracle_bit_exact(model, k, hrs):
    w, h = 384, 216
    lens = S.gopro_style_lens(w, h)
    lens["model"] = model
    lens["k"] = list(k) + [0.0] * (12 - len(k))
    fr = S.SyntheticFrame("YUV422P16LE", w, h, seed=5, lens=lens, horizontal_rs=hrs, fov=1.4)
    kp = fr.planes[0]["params"].copy()
    kp.flags = abi.FLAG_HORIZONTAL_RS if hrs else 0          # stmap.rs:36-38
    ref = O.stmap_undistort(kp, fr.model, 0, fr.matrices, w, h)
    pl = fr.planes[0]
    b = warp.host_buffers(pl["src"], pl["size"], pl["dst"].copy(), pl["out_size"])
    be = warp.Backend(pl["params"], pl["pixel_type"], fr.model, 0, b)
    try:
        got = be.stmap_undistort(kp, fr.matrices, w, h)
    finally:
        be.close()
    assert np.array_equal(ref.view(np.uint32), got.view(np.uint32))
    assert np.count_nonzero(ref) > 0.5 * ref.size
