#!/bin/bash
# tools/gpu_call.sh <tag> <script-body-file>: one GPU call = one tagged directory under gpurun_out/ with a log (replaces the per-call scripts of rounds 3-4).
# Helpers for the body: bench <env> <args...> appends one line (value Mpix/s, ms/step, kernel ms/frame, compile ms, parity) to $O/summary.txt.
cd $GRAFT_REPO_ROOT
TAG=${1:?tag}; BODY=${2:?body}
O=gpurun_out/$TAG; mkdir -p $O; export O
bench() { local e="$1"; shift; env $e timeout 300 python3 bench.py --gpus 1 --no-cpu-baseline "$@" > $O/bench.json 2> $O/bench.err
  python3 -c "import json; d=json.load(open('$O/bench.json')); print('[$e] [$*]', d['value'], d['ms_per_step'], d.get('roofline', {}).get('kernel_ms_per_frame'), d['config']['jit']['compile_ms'], d['config'].get('parity_vs_oracle'))" 2>&1 | tail -1 | tee -a $O/summary.txt; }
source $BODY 2>&1 | tee -a $O/log.txt
