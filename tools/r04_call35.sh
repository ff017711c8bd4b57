#!/bin/bash
# round 4, GPU call 35: hiprtc before / after a HIP device is initialised in the process (the shipped kernel cache is compiled device-less)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04zi; mkdir -p $O
timeout 200 python3 tools/diag_hiprtc_device.py $O 2>&1 | tail -8 | tee $O/summary.txt
ls -la $O | tee -a $O/summary.txt
