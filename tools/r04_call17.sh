#!/bin/bash
# round 4, GPU call 17: what deriving E in the kernel costs (C2; GFW_P1_BOUND_OFF = a constant E, A/B only)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04q; mkdir -p $O
for v in "" "GFW_P1_BOUND_OFF=1" "" "GFW_P1_BOUND_OFF=1" ""; do
  GFW_JIT_DEFS="$v" timeout 300 python3 bench.py --gpus 1 --steps 200 --warmup 20 --no-cpu-baseline > $O/bench.json 2> $O/bench.err
  python3 -c "import json; d=json.load(open('$O/bench.json')); print('[$v]',d['value'], d['ms_per_step'], d['roofline']['kernel_ms_per_frame'], d['config']['backend'], d['config']['parity_vs_oracle'])" 2>&1 | tail -1 | tee -a $O/summary.txt
done
timeout 300 python -m pytest tests/test_gpu_pass1.py tests/test_gpu_pass1_sweep.py -m gpu -q -p no:cacheprovider 2>&1 | tail -3 | tee -a $O/summary.txt
