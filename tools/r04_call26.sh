#!/bin/bash
# round 4, GPU call 26: Lanczos4 / bicubic — waves per SIMD against tap rows in flight (the 8-wave budget of 64 VGPRs holds two rows; 6 waves hold four)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04z; mkdir -p $O
run() { GFW_JIT_WAVES=$1 GFW_JIT_DEFS="GFW_TAP_ROW_UNROLL=$2" timeout 300 python3 bench.py --gpus 1 --steps 64 --warmup 16 --no-cpu-baseline --interp $3 > $O/bench.json 2> $O/bench.err
  python3 -c "import json; d=json.load(open('$O/bench.json')); print('[waves $1 rows $2 taps $3]',d['value'], d['ms_per_step'], d['roofline']['kernel_ms_per_frame'], d['config']['backend'], d['config']['parity_vs_oracle'])" 2>&1 | tail -1 | tee -a $O/summary.txt; }
run 8 2 8
run 7 2 8
run 6 2 8
run 6 4 8
run 7 4 8
run 7 2 4
run 6 4 4
run 8 2 4
