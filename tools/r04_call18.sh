#!/bin/bash
# round 4, GPU call 18: where the cost of the in-kernel E sits (C2): launch start or frame change
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04r; mkdir -p $O
run() { GFW_JIT_DEFS="$1" timeout 300 python3 bench.py --gpus 1 --steps 200 --warmup 20 --no-cpu-baseline $2 > $O/bench.json 2> $O/bench.err
  python3 -c "import json; d=json.load(open('$O/bench.json')); print('[$1] [$2]',d['value'], d['ms_per_step'], d['roofline']['kernel_ms_per_frame'], d['config']['backend'], d['config']['parity_vs_oracle'])" 2>&1 | tail -1 | tee -a $O/summary.txt; }
run "" ""
run "" ""
run "GFW_P1_BOUND_OFF=1" ""
run "" "--clip 1"
run "GFW_P1_BOUND_OFF=1" "--clip 1"
run "" "--clip 16"
run "GFW_P1_BOUND_OFF=1" "--clip 16"
run "" ""
