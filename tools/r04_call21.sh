#!/bin/bash
# round 4, GPU call 21: per-wave timelines of the specialised kernel: one frame per launch against eight (where do the 10 us per launch go?)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04u; mkdir -p $O
run() { GFW_TIMELINE_FILE=$O/tl_$1.bin GFW_JIT_DEFS="GFW_TIMELINE=1" timeout 300 python3 bench.py --gpus 1 --steps 200 --warmup 20 --no-cpu-baseline $2 > $O/bench.json 2> $O/bench.err
  python3 -c "import json; d=json.load(open('$O/bench.json')); print('[$1] [$2]',d['value'], d['ms_per_step'], d['roofline']['kernel_ms_per_frame'], d['config']['backend'], d['config']['parity_vs_oracle'])" 2>&1 | tail -1 | tee -a $O/summary.txt
  python3 tools/analyze_timeline.py $O/tl_$1.bin 2048 2>&1 | tee -a $O/summary.txt; }
run clip1 "--clip 1"
run clip8 ""
