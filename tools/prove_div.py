#!/usr/bin/env python3
"""Exhaustive machine check of the kernels' divide sequences on the device (run on an MI355X):

    python tools/prove_div.py [test_id] [chunks]

For every pair of f32 significands (2^23 x 2^23 = 7.04e13 quotients) the sequence is compared with the generic
correctly-rounded `a / b`.  test 3 = refined reciprocal + one remainder correction (what gfw_div_prepared uses),
4 = raw v_rcp_f32 + two corrections, 5 = raw v_rcp_f32 + one correction (expected to fail: shown for contrast,
subsampled).  Prints one line per chunk of 2^20 denominators and the total number of mismatching quotients."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gyroflow_amd import abi  # noqa: E402

lib = abi.load_library()
test = int(sys.argv[1]) if len(sys.argv) > 1 else 3
chunks = int(sys.argv[2]) if len(sys.argv) > 2 else 8
log2_step = int(sys.argv[3]) if len(sys.argv) > 3 else 0
per = (1 << 23) // chunks
total = 0
t0 = time.time()
for c in range(chunks):
    bad = lib.gfw_debug_selftest(test, per, (log2_step << 32) | (c * per))
    if bad < 0:
        raise SystemExit("error %d: %s" % (bad, lib.gfw_last_error().decode()))
    total += bad
    print("test %d  denominators [%d, %d)  numerator step %d  mismatches %d  (%.1f s)" % (test, c * per, (c + 1) * per, 1 << log2_step, bad, time.time() - t0), flush=True)
print("TOTAL test %d: %d mismatching quotients out of %.3e" % (test, total, float(1 << 23) * ((1 << 23) >> log2_step)))
