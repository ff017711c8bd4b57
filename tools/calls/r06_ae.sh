#!/bin/bash
# round 6, call ae: the paired-EWA test with the four-plane formats added
O=gpurun_out/r06_ae; mkdir -p $O; : > $O/summary.txt
export TMPDIR=/tmp
timeout 900 python3 -m pytest tests/test_gpu_ewa_pair.py -q -m gpu --tb=short -p no:cacheprovider 2>&1 | grep -v "amdgpu.ids" | tail -15 | tee -a $O/summary.txt
