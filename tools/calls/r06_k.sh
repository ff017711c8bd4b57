#!/bin/bash
O=gpurun_out/r06_k; mkdir -p $O; : > $O/summary.txt
export TMPDIR=/tmp
timeout 600 python3 tools/diag_clip_scratch.py 2>&1 | grep -v amdgpu.ids | tee -a $O/summary.txt
