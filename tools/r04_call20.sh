#!/bin/bash
# round 4, GPU call 20: the launch-shape knobs again, after the row became branch-free (C2): frames per launch, priority feedback
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04t; mkdir -p $O
run() { GFW_JIT_DEFS="$1" timeout 300 python3 bench.py --gpus 1 --steps 240 --warmup 24 --no-cpu-baseline $2 > $O/bench.json 2> $O/bench.err
  python3 -c "import json; d=json.load(open('$O/bench.json')); print('[$1] [$2]',d['value'], d['ms_per_step'], d['roofline']['kernel_ms_per_frame'], d['config']['backend'], d['config']['parity_vs_oracle'])" 2>&1 | tail -1 | tee -a $O/summary.txt; }
run "" ""
run "" "--clip 4"
run "" "--clip 6"
run "" "--clip 10"
run "" "--clip 12"
run "GFW_PRIO_MODE=0" ""
run "GFW_PRIO_SPAN=4" ""
run "GFW_PRIO_SPAN=10" ""
run "" "--jit 0 --clip 1"
run "" "--per-plane --clip 1"
