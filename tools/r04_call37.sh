#!/bin/bash
# round 4, GPU call 37 (last): comgr's own log of the compile the run-time path performs inside the benchmark process (to compare its options with the build step's)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04zk; mkdir -p $O
AMD_COMGR_EMIT_VERBOSE_LOGS=1 AMD_COMGR_REDIRECT_LOGS=$O/comgr_live.log GFW_JIT_DEFS=GFW_UNUSED_TAG=9 timeout 100 python3 bench.py --gpus 1 --steps 32 --warmup 8 --no-cpu-baseline --no-parity --fmt P010LE --interp 4 > $O/bench.json 2> $O/bench.err
python3 -c "import json; d=json.load(open('$O/bench.json')); print(d['value'], d['ms_per_step'], d['config']['jit']['compile_ms'])" | tee $O/summary.txt
grep -c . $O/comgr_live.log
