#!/bin/bash
# round 6, call aa: the closed-form sweep's "served" count (14 of 20 once in the serial suite of r06_final4; 15 asked): three runs alone, output kept
O=gpurun_out/r06_aa; mkdir -p $O; : > $O/summary.txt
export TMPDIR=/tmp
for i in 1 2 3; do timeout 900 python3 -m pytest "tests/test_gpu_pass1_radial.py::test_twenty_random_closed_form_clips_never_produce_a_wrong_certificate" -q -m gpu -s --tb=short -p no:cacheprovider 2>&1 | grep -v "amdgpu.ids" | grep "certified first pass\|passed\|failed\|Error" | tee -a $O/summary.txt; done
