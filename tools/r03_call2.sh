#!/bin/bash
# round 3, second GPU call: staged paths through the parity suite, reference-OpenCL residual calibration, Lanczos4 LDS tile, baked constants
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03b
GFW_LIBRARY=$GRAFT_REPO_ROOT/variants/libgfwarp_staged.so timeout 600 python -m pytest tests -m gpu_staged -q -p no:cacheprovider > gpurun_out/r03b/staged.log 2>&1
tail -5 gpurun_out/r03b/staged.log
timeout 600 python tools/ref_residual.py > gpurun_out/r03b/ref_residual.log 2>&1
tail -40 gpurun_out/r03b/ref_residual.log
bash tools/gpu_ab.sh r03b base bake6 "bake8:--grid 2048" "bake8a:--grid 2048" "bake7a:--grid 1792" "atan_w8:--grid 2048" "bake8a:--grid 2048 --streams 2" base \
   "l8_base:--interp 8 --steps 60" "l8_tile:--interp 8 --steps 60" "l8_tile:--interp 8 --steps 60 --grid 1024" "l8_tilef:--interp 8 --steps 60 --grid 768"
