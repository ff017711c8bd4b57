#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02h
for v in 0 4; do for g in 1536 1352 1016 2032 4056 816; do
timeout 300 python3 bench.py --gpus 1 --steps 200 --warmup 20 --no-cpu-baseline --no-parity --variant $v --grid $g > gpurun_out/r02h/bench_v${v}_g$g.json 2> gpurun_out/r02h/bench_v${v}_g$g.err
python3 -c "import json; d=json.load(open('gpurun_out/r02h/bench_v${v}_g$g.json')); print('variant',$v,'grid',$g,d['value'], round(d['ms_per_step']*1000,2), round(d['roofline']['kernel_ms_per_launch']*1000,2), d['config']['backend'])" | tee -a gpurun_out/r02h/summary.txt
done; done
