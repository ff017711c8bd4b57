#!/bin/bash
# round 4, GPU call 32: C2 bicubic, one binary, two rates (78 us from the shipped cache, 64 us compiled on the box): which one is sustained?
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04zf; mkdir -p $O
run() { env $1 timeout 300 python3 bench.py --gpus 1 --warmup 16 --no-cpu-baseline --interp 4 $2 > $O/bench.json 2> $O/bench.err
  python3 -c "import json; d=json.load(open('$O/bench.json')); print('[$1] [$2]',d['value'], d['ms_per_step'], d['roofline']['kernel_ms_per_frame'], d['config']['jit']['compile_ms'], d['config']['parity_vs_oracle'])" 2>&1 | tail -1 | tee -a $O/summary.txt; }
run "A=1" "--steps 1024"
run "GFW_JIT_DEFS=GFW_UNUSED_TAG=1" "--steps 1024"
run "A=1" "--steps 64 --preheat-ms 600"
run "GFW_JIT_WAVES=8 GFW_JIT_DEFS=GFW_TAP_ROWS_FORCE=2" "--steps 1024"
