#!/bin/bash
# round 3, twenty-ninth GPU call (the last 48 GPU-seconds): the clip-launch fix for background mode 3 on planar chroma — the GPU twin of the interpreter's finding
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03v; mkdir -p $O
timeout 40 python -m pytest tests/test_gpu_jit.py -m gpu -q -p no:cacheprovider -x -k "generic_model_bodies" > $O/clip_bg3.log 2>&1; echo "pytest rc $?"; tail -4 $O/clip_bg3.log
