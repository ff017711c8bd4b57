#!/bin/bash
# round 4, GPU call 24: C4 (packed RGBAf crop) through this round's earlier libraries on one box: which change cost 68 -> 72 us (or is it the box)?
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04x; mkdir -p $O
run() { lib=$1; shift; L=""; [ "$lib" != head ] && L=$GRAFT_REPO_ROOT/variants/libgfwarp_$lib.so
  GFW_LIBRARY=$L timeout 300 python3 bench.py --gpus 1 --steps 128 --warmup 16 --no-cpu-baseline --resident 16 "$@" > $O/bench.json 2> $O/bench.err
  python3 -c "import json; d=json.load(open('$O/bench.json')); print('[$lib] [$*]',d['value'], d['ms_per_step'], d['roofline']['kernel_ms_per_frame'], d['config']['backend'], d['config']['parity_vs_oracle'])" 2>&1 | tail -1 | tee -a $O/summary.txt; }
for l in head e98a0e1 f75c3ed 16e3b9e head e98a0e1; do run $l --fmt RGBAF32 --crop; done
for l in head e98a0e1; do run $l --fmt P010LE; done
