#!/bin/bash
# round 4, GPU call 19: per-frame E table in LDS: single-frame launches, first-pass audits, the GPU tier, the driver command
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04s; mkdir -p $O
run() { GFW_JIT_DEFS="$1" timeout 300 python3 bench.py --gpus 1 --steps 200 --warmup 20 --no-cpu-baseline $2 > $O/bench.json 2> $O/bench.err
  python3 -c "import json; d=json.load(open('$O/bench.json')); print('[$1] [$2]',d['value'], d['ms_per_step'], d['roofline']['kernel_ms_per_frame'], d['config']['backend'], d['config']['parity_vs_oracle'])" 2>&1 | tail -1 | tee -a $O/summary.txt; }
run "" "--clip 1"
run "GFW_P1_BOUND_OFF=1" "--clip 1"
run "" "--jit 0"
timeout 1500 python -m pytest tests -m gpu -q -x -p no:cacheprovider > $O/gpu_tests.log 2>&1; echo "gpu tests rc $?" | tee -a $O/summary.txt; tail -5 $O/gpu_tests.log
timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver.json 2> $O/bench_driver.err; python3 -c "
import json; d=json.load(open('$O/bench_driver.json')); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['kernel_ms_per_frame'], d['cpu_baseline']['value'], d['cpu_baseline']['cores'], d['config']['parity_vs_oracle'])" | tee -a $O/summary.txt
