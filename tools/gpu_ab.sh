#!/bin/bash
# tools/gpu_ab.sh <outdir> <spec>...   spec = variant[:bench args]  — A/B bench of variants/libgfwarp_<variant>.so on the GPU box
cd $GRAFT_REPO_ROOT
O=gpurun_out/$1; shift; mkdir -p $O
i=0
for spec in "$@"; do
  name=${spec%%:*}; args=""; [[ "$spec" == *:* ]] && args=${spec#*:}
  i=$((i+1)); tag=${i}_${name}
  GFW_TIMELINE_FILE=$O/tl_$tag.bin GFW_LIBRARY=$GRAFT_REPO_ROOT/variants/libgfwarp_$name.so timeout 300 python3 bench.py --gpus 1 --steps 200 --warmup 20 --no-cpu-baseline $args > $O/bench_$tag.json 2> $O/bench_$tag.err
  python3 -c "import json; d=json.load(open('$O/bench_$tag.json')); print('$name [$args]',d['value'], d['ms_per_step'], d['roofline']['kernel_ms_per_launch'], d['config']['backend'], d['config']['parity_vs_oracle'])" 2>&1 | tail -1 | tee -a $O/summary.txt
done
