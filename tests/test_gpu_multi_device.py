"""N renders on N threads of ONE process through the C ABI, each thread on its own device (gfw_set_device per thread) with its own context
(tests/cpp/test_multi_device.cpp; round-5 verdict, next #9).  The reference's closest shape to multi-GPU is exactly this — `--parallel-renders`, several renders side by
side in one process (src/cli.rs:46-48, render_queue.rs:677), backends in thread-local caches (stabilization/mod.rs:59-66).  Frames are dealt round-robin;
every frame's checksum, taken where the pixels leave (gfw_set_frame_checksums), must equal the one-thread run's.  One GPU here: four threads share device 0; on an
8-GPU node the same binary puts thread t on device t mod 8."""
import os
import subprocess

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "tests", "cpp", "test_multi_device")


@pytest.mark.parametrize("threads,frames", [(4, 36), (8, 40)])
def test_threads_with_their_own_device_and_context_reproduce_the_one_thread_checksums(threads, frames):
    if not os.path.exists(EXE):
        pytest.skip("tests/cpp/test_multi_device not built (__graft_entry__.build)")
    r = subprocess.run([EXE, str(threads), str(frames)], capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, "rc %s\n" % r.returncode + r.stdout[-3000:] + r.stderr[-3000:]
    assert "multi-device ok" in r.stdout and "frames whose checksum differs: 0; zero checksums: 0" in r.stdout, r.stdout
    # both kinds of kernel served frames of the one-thread run: ahead of time first, the run-time specialised one once its background build landed
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("multi-device:")][0]
    assert "yuv_fused" in line, line


def test_a_process_may_leave_main_while_its_first_build_is_compiling():
    """GFW_OPT_JIT = 1 (the default) compiles a clip's kernel on a worker thread; a short render can be over, its contexts destroyed and main() left while that build is
    still inside hiprtc.  The library's exit hook joins it — and must be OLDER than the compiler's own exit handlers, which hiprtc would otherwise register lazily at
    the first compile (one run in thirty of the multi-device test hung at exit beside other GPU processes before round 6 loaded the compiler up front: gfw_jit.hip)."""
    if not os.path.exists(EXE):
        pytest.skip("tests/cpp/test_multi_device not built (__graft_entry__.build)")
    for rep in range(6):
        r = subprocess.run([EXE, "exit"], capture_output=True, text=True, timeout=180, cwd=ROOT, env=dict(os.environ, GFW_JIT_CACHE=""))
        assert r.returncode == 0 and "exit-during-build: leaving main" in r.stdout, "repetition %d: rc %s\n" % (rep, r.returncode) + r.stdout[-2000:] + r.stderr[-2000:]
