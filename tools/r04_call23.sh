#!/bin/bash
# round 4, GPU call 23: C4 (packed RGBAf, crop) — is the 68 -> 72 us the certificate's width, its evaluation, or neither?
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04w; mkdir -p $O
run() { GFW_JIT_DEFS="$1" timeout 300 python3 bench.py --gpus 1 --steps 128 --warmup 16 --no-cpu-baseline --fmt RGBAF32 --crop --resident 16 $2 > $O/bench.json 2> $O/bench.err
  python3 -c "import json; d=json.load(open('$O/bench.json')); print('[$1] [$2]',d['value'], d['ms_per_step'], d['roofline']['kernel_ms_per_frame'], d['config']['backend'], d['config']['parity_vs_oracle'])" 2>&1 | tail -1 | tee -a $O/summary.txt; }
run "" ""
run "GFW_P1_E_SCALE=0.5f" ""
run "GFW_P1_BOUND_OFF=1" ""
run "" ""
run "GFW_FASTROW=0" ""
