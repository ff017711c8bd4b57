#!/bin/bash
# Round 5, first GPU call (queued at the end of round 4, whose GPU budget ended before these could be timed): the LUT samplers after the group-first tap-row fetch
# and the merged last-dword region, through the SHIPPED kernel cache (ROCm 7.2) and, with a tag that forces a compile, through the hiprtc torch brings (ROCm 7.0);
# C2 bicubic at six waves / four rows (78 us from 7.2 before the change, 64 from 7.0: under the 65 us mark if the change fixed the 7.2 build as it fixed NV12's).
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05a; mkdir -p $O
run() { env $1 timeout 120 python3 bench.py --gpus 1 --steps 64 --warmup 16 --no-cpu-baseline $2 $3 $4 $5 > $O/bench.json 2> $O/bench.err
  python3 -c "import json; d=json.load(open('$O/bench.json')); print('[$1] [$2 $3 $4 $5]',d['value'], d['ms_per_step'], d['roofline']['kernel_ms_per_frame'], d['config']['jit']['compile_ms'], d['config']['parity_vs_oracle'])" 2>&1 | tail -1 | tee -a $O/summary.txt; }
# C2 row with its fetches in one cluster (GFW_ROW_CLUSTER): both sides compiled in the process (the same compiler: torch's hiprtc), then the shipped default for reference
run GFW_JIT_DEFS=GFW_UNUSED_TAG=1 --steps 200
run GFW_JIT_DEFS=GFW_ROW_CLUSTER=1 --steps 200
run GFW_JIT_DEFS=GFW_UNUSED_TAG=1 --steps 200
run GFW_JIT_DEFS=GFW_ROW_CLUSTER=1 --steps 200
run A=1 --steps 200
run A=1 --interp 4
run GFW_JIT_WAVES=6 --interp 4
run "GFW_JIT_WAVES=6 GFW_JIT_DEFS=GFW_UNUSED_TAG=1" --interp 4
run A=1 --interp 8
run A=1 --fmt NV12 --interp 4
run A=1 --fmt NV12 --interp 8
run A=1 --fmt P010LE --interp 4
run A=1 --fmt P010LE --interp 8
run A=1 --fmt YUV420P --interp 8
run A=1 --interp 4 --jit 0 --clip 1
run A=1 --interp 8 --jit 0 --clip 1
run A=1 --fmt GBRAPF32LE --crop --resident 16
run A=1 --fmt RGBAF32 --crop --resident 16
run A=1 --fmt YUV444P16LE
