#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02l
for name in base pair dot base; do
GFW_LIBRARY=$GRAFT_REPO_ROOT/variants/libgfwarp_$name.so timeout 300 python3 bench.py --gpus 1 --steps 200 --warmup 20 --no-cpu-baseline > gpurun_out/r02l/bench_$name.json 2> gpurun_out/r02l/bench_$name.err
python3 -c "import json; d=json.load(open('gpurun_out/r02l/bench_$name.json')); print('$name',d['value'], d['ms_per_step'], d['roofline']['kernel_ms_per_launch'], d['config']['backend'], d['config']['parity_vs_oracle'])" | tee -a gpurun_out/r02l/summary.txt
done
