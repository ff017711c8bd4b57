#!/bin/bash
# round 4, GPU call 28: LUT samplers — waves per SIMD against tap rows in flight, third scan (8-bit Lanczos4 at 4 waves, the other 4:2:0 formats)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04zb; mkdir -p $O
run() { GFW_JIT_WAVES=$1 GFW_JIT_DEFS="GFW_TAP_ROW_UNROLL=$2" timeout 300 python3 bench.py --gpus 1 --steps 64 --warmup 16 --no-cpu-baseline --interp $3 $4 $5 > $O/bench.json 2> $O/bench.err
  python3 -c "import json; d=json.load(open('$O/bench.json')); print('[waves $1 rows $2 taps $3 $4 $5]',d['value'], d['ms_per_step'], d['roofline']['kernel_ms_per_frame'], d['config']['backend'], d['config']['parity_vs_oracle'])" 2>&1 | tail -1 | tee -a $O/summary.txt; }
run 4 8 8 --fmt NV12
run 5 8 8 --fmt NV12
run 5 4 4 --fmt NV12
run 8 2 8 --fmt YUV420P
run 5 8 8 --fmt YUV420P
run 8 2 4 --fmt P010LE
run 6 4 4 --fmt P010LE
run 8 2 8 --fmt P010LE
run 6 4 8 --fmt P010LE
