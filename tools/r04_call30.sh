#!/bin/bash
# round 4, GPU call 30: C2 bicubic with the new defaults — 78 us in call 29 against 64 in the scans: position in the call, the shipped cache, or the definitions?
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04zd; mkdir -p $O
run() { env $1 timeout 300 python3 bench.py --gpus 1 --steps 64 --warmup 16 --no-cpu-baseline --interp 4 > $O/bench.json 2> $O/bench.err
  python3 -c "import json; d=json.load(open('$O/bench.json')); print('[$1]',d['value'], d['ms_per_step'], d['roofline']['kernel_ms_per_frame'], d['config']['backend'], d['config']['jit'], d['config']['parity_vs_oracle'])" 2>&1 | tail -1 | tee -a $O/summary.txt; }
run "A=1"
run "A=2"
run "GFW_JIT_DEFS=GFW_TAP_ROWS_FORCE=4"
run "GFW_JIT_WAVES=6"
run "GFW_JIT_CACHE=/nonexistent_dir_for_cache GFW_JIT_DEFS=X=1"
run "A=3"
