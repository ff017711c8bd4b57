#!/bin/bash
# round 4, GPU call 34: C2 bicubic with the shipped kernel cache hidden: what does the run-time path itself compile, under which name, how fast?
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04zh; mkdir -p $O $O/cache2
mv gyroflow_amd/jit_cache gyroflow_amd/jit_cache_hidden
run() { env $1 timeout 300 python3 bench.py --gpus 1 --steps 64 --warmup 16 --no-cpu-baseline --interp 4 > $O/bench.json 2> $O/bench.err
  python3 -c "import json; d=json.load(open('$O/bench.json')); print('[$1]',d['value'], d['ms_per_step'], d['roofline']['kernel_ms_per_frame'], d['config']['jit']['compile_ms'], d['config']['parity_vs_oracle'])" 2>&1 | tail -1 | tee -a $O/summary.txt; }
run "GFW_JIT_CACHE=$O/cache2"
run "GFW_JIT_CACHE=$O/cache2"
mv gyroflow_amd/jit_cache_hidden gyroflow_amd/jit_cache
ls -la $O/cache2 | tee -a $O/summary.txt
