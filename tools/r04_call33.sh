#!/bin/bash
# round 4, GPU call 33: C2 bicubic from the kernel cache after a hiprtc build in the same process / a kernel cache written by an earlier process
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04zg; mkdir -p $O $O/cache
run() { env $1 timeout 300 python3 $2 --gpus 1 --steps 64 --warmup 16 --no-cpu-baseline --interp 4 > $O/bench.json 2> $O/bench.err
  python3 -c "import json; d=json.load(open('$O/bench.json')); print('[$1] [$2]',d['value'], d['ms_per_step'], d['roofline']['kernel_ms_per_frame'], d['config']['jit']['compile_ms'], d['config']['parity_vs_oracle'])" 2>&1 | tail -1 | tee -a $O/summary.txt; }
run "A=1" "tools/bench_with_prelude.py compile"
run "A=1" "tools/bench_with_prelude.py none"
run "GFW_JIT_CACHE=$O/cache GFW_JIT_DEFS=GFW_UNUSED_TAG=5" "bench.py"
run "GFW_JIT_CACHE=$O/cache GFW_JIT_DEFS=GFW_UNUSED_TAG=5" "bench.py"
