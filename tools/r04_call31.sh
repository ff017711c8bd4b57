#!/bin/bash
# round 4, GPU call 31: is the box's hiprtc output for C2 bicubic the binary the build step shipped?  And the same definitions compiled on the box, timed
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04ze; mkdir -p $O
python3 tools/cmp_cache_entry.py $O/box_bicubic.co 2>&1 | tail -1 | tee -a $O/summary.txt
run() { env $1 timeout 300 python3 bench.py --gpus 1 --steps 64 --warmup 16 --no-cpu-baseline --interp 4 > $O/bench.json 2> $O/bench.err
  python3 -c "import json; d=json.load(open('$O/bench.json')); print('[$1]',d['value'], d['ms_per_step'], d['roofline']['kernel_ms_per_frame'], d['config']['backend'], d['config']['jit']['compile_ms'], d['config']['parity_vs_oracle'])" 2>&1 | tail -1 | tee -a $O/summary.txt; }
run "GFW_JIT_DEFS=GFW_UNUSED_TAG=1"
run "A=1"
run "GFW_JIT_DEFS=GFW_UNUSED_TAG=2"
