"""A second, independent statement for branches of the CPU path that no reference code can pin here (the OpenCL twin differs from them by design, the Rust
cannot be built): the colour-range fix (cpu_undistort.rs:254-260 — `px *= 0.85882352 | 0.87843137; px[0] += 16; px[1] += 16` before the store, applied to
background pixels too, :615-621) and the three-channel pixel types' load / store (pixel_formats.rs).  Written in numpy float32 from the Rust, not from
oracle/gfw_oracle.c: on a plane of ONE colour every bilinear sample is that colour exactly (tap x weight products and their sums are exact for 8-bit values and
the k/32 weights), so the expected output is a closed form — sampled pixels remapped, out-of-frame pixels = remapped background — and any disagreement is the
oracle's.  The GPU tier holds libgfwarp to the oracle on the same flag (tests/test_gpu_parity.py::test_fix_color_range_flag)."""
import numpy as np
import pytest

from gyroflow_amd import abi, synthetic as S
import _oracle as O


def remap_colorrange(px, is_y):
    """cpu_undistort.rs:254-260 in float32, one rounding per operation"""
    px = np.asarray(px, dtype=np.float32) * (np.float32(0.85882352) if is_y else np.float32(0.87843137))
    px[..., 0] = px[..., 0] + np.float32(16.0)
    if px.shape[-1] > 1:
        px[..., 1] = px[..., 1] + np.float32(16.0)
    return px


def as_u8(v):
    """Rust `as u8`: truncate, saturate, NaN -> 0"""
    return np.clip(np.trunc(np.nan_to_num(v, nan=0.0)), 0, 255).astype(np.uint8)


def constant_frame(fmt, colours, flags, bg):
    fr = S.SyntheticFrame(fmt, 160, 96, seed=4, fov=2.0, flags=flags, background_rgba=bg)         # fov 3: much of the output lies outside the source
    for pl, col in zip(fr.planes, colours):
        h, stride = pl["size"][1], pl["size"][2]
        n = len(col)
        row = np.zeros(stride, np.uint8)
        row[:pl["size"][0] * n] = np.tile(np.array(col, np.uint8), pl["size"][0])
        pl["src"][:] = np.tile(row, h)
    return fr


@pytest.mark.parametrize("fix", [False, True])
def test_colour_range_fix_on_planes_of_one_colour(fix):
    flags = abi.FLAG_FIX_COLOR_RANGE if fix else 0
    fr = constant_frame("NV12", [(200,), (90, 170)], flags, (0.25, 0.5, 0.75, 1.0))
    outs = O.run_frame(fr)
    for p, (pl, out, col) in enumerate(zip(fr.planes, outs, [(200,), (90, 170)])):
        w, h, stride = pl["out_size"]
        n = len(col)
        prm = pl["params"]
        img = out.reshape(h, stride)[:, :w * n].reshape(h, w, n)
        sampled = np.array(col, np.float32)
        bgv = np.array([np.float32(prm.background[c]) * np.float32(prm.max_pixel_value) for c in range(n)], np.float32)     # :523
        want_in = as_u8(remap_colorrange(sampled[None, :], p == 0)[0] if fix else sampled)
        want_bg = as_u8(remap_colorrange(bgv[None, :], p == 0)[0] if fix else bgv)
        flat = img.reshape(-1, n)
        is_in = np.all(flat == want_in, axis=1)
        is_bg = np.all(flat == want_bg, axis=1)
        # every pixel is one of the two closed forms, except the thin band where taps straddle the frame edge (a blend of both); both kinds occur
        assert is_in.sum() > 0.3 * len(flat) and is_bg.sum() > 0.02 * len(flat), (p, is_in.sum(), is_bg.sum(), len(flat))
        assert (is_in | is_bg).sum() > 0.9 * len(flat), (p, (is_in | is_bg).sum(), len(flat))
        if fix and p == 0:
            assert int(want_in[0]) == int(np.trunc(np.float32(200) * np.float32(0.85882352) + np.float32(16.0)))


def test_three_channel_pixels_keep_their_channels():
    """RGB8 (the OpenCL backend's own FIXME: pixel_formats.rs:114): load / store of three interleaved channels — each keeps its place and value"""
    fr = constant_frame("RGB24", [(10, 130, 250)], 0, (0.2, 0.4, 0.6, 1.0))
    out = O.run_frame(fr)[0]
    w, h, stride = fr.planes[0]["out_size"]
    img = out.reshape(h, stride)[:, :w * 3].reshape(-1, 3)
    prm = fr.planes[0]["params"]
    bgv = as_u8(np.array([np.float32(prm.background[c]) * np.float32(prm.max_pixel_value) for c in range(3)], np.float32))
    is_in = np.all(img == np.array([10, 130, 250], np.uint8), axis=1)
    is_bg = np.all(img == bgv, axis=1)
    assert is_in.sum() > 0.3 * len(img) and is_bg.sum() > 0.02 * len(img) and (is_in | is_bg).sum() > 0.9 * len(img)
