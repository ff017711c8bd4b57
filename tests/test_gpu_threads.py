"""Concurrency contract (SURVEY.md 8b "Threading"): `Stabilization` is Send + Sync and `process_pixels` runs from several
threads at once, each thread owning its backend object (thread-local caches, mod.rs:59-66).  Here four threads, each with
its own context and stream, warp different frames concurrently (ctypes releases the GIL during the calls); every result
must equal the oracle's."""
import threading

import numpy as np
import pytest

from gyroflow_amd import synthetic as S, warp
import _oracle as O

pytestmark = pytest.mark.gpu


def test_four_threads_with_their_own_contexts():
    cases = [("YUV422P16LE", 320, 180, 2), ("NV12", 256, 144, 4), ("RGBA", 200, 120, 8), ("YUV420P10LE", 192, 128, 11)]
    frames = [S.SyntheticFrame(fmt, w, h, seed=200 + i, interpolation=interp, fov=1.2) for i, (fmt, w, h, interp) in enumerate(cases)]
    refs = [O.run_frame(fr) for fr in frames]
    errors, results = [], [None] * len(frames)

    def worker(i):
        try:
            fr = frames[i]
            outs = None
            for _ in range(12):                       # several calls per thread so that the threads really overlap
                outs = warp.run_frame(fr)
            results[i] = outs
        except Exception as e:                        # noqa: BLE001 - reported below
            errors.append((i, repr(e)))

    threads = [threading.Thread(target=worker, args=(i,)) for i in range(len(frames))]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=300)
    assert not errors, errors
    for i, (ref, got) in enumerate(zip(refs, results)):
        assert got is not None
        for a, b in zip(ref, got):
            assert np.array_equal(a, b), "thread %d (%s)" % (i, cases[i][0])
