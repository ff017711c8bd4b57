#!/bin/bash
# round 4, GPU call 36 (the round's last minutes): the LUT samplers through the SHIPPED kernel cache (compiled by the build step) with the final rules
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04zj; mkdir -p $O
run() { timeout 120 python3 bench.py --gpus 1 --steps 64 --warmup 16 --no-cpu-baseline --no-parity $1 $2 $3 $4 > $O/bench.json 2> $O/bench.err
  python3 -c "import json; d=json.load(open('$O/bench.json')); print('[$1 $2 $3 $4]',d['value'], d['ms_per_step'], d['roofline']['kernel_ms_per_frame'], d['config']['jit']['compile_ms'])" 2>&1 | tail -1 | tee -a $O/summary.txt; }
run --interp 4
run --fmt NV12 --interp 4
run --fmt NV12 --interp 8
run --fmt P010LE --interp 4
timeout 60 python -m pytest tests/test_gpu_jit_cache.py -m gpu -q -x -p no:cacheprovider 2>&1 | tail -2 | tee -a $O/summary.txt
