#!/bin/bash
# round 6, call i: where the closed-form radial first pass differs from the oracle on the GPU (diagnosis), hazard scan of the kernels it compiled
O=gpurun_out/r06_i; mkdir -p $O; : > $O/summary.txt
export TMPDIR=/tmp GFW_JIT_CACHE=/tmp/jitc; mkdir -p /tmp/jitc
timeout 900 python3 tools/diag_radial.py 2>&1 | grep -v amdgpu.ids | tee -a $O/summary.txt
ls /tmp/jitc | head -40 >> $O/summary.txt
timeout 600 python3 tools/scan_trans_hazard.py /tmp/jitc/* 2>&1 | tail -30 | tee -a $O/summary.txt
mkdir -p $O/jitc; cp /tmp/jitc/* $O/jitc/ 2>/dev/null; du -sh $O/jitc | tee -a $O/summary.txt
