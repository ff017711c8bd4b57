#!/bin/bash
# round 4, GPU call 38 (the last seconds of the budget): tap rows fetched group-first, through the shipped (ROCm 7.2) kernel cache, parity on
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04zl; mkdir -p $O
run() { timeout 25 python3 bench.py --gpus 1 --steps 64 --warmup 16 --no-cpu-baseline $1 $2 $3 $4 > $O/bench.json 2> $O/bench.err
  python3 -c "import json; d=json.load(open('$O/bench.json')); print('[$1 $2 $3 $4]',d['value'], d['ms_per_step'], d['roofline']['kernel_ms_per_frame'], d['config']['jit']['compile_ms'], d['config']['parity_vs_oracle'])" 2>&1 | tail -1 | tee -a $O/summary.txt; }
run --fmt NV12 --interp 4
run --interp 4
