#!/bin/bash
# round 4, GPU call 16: what the width of the derived certificate costs (C2, run-time specialised kernel, E scaled through GFW_JIT_DEFS)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04p; mkdir -p $O
for sc in 1.0f 0.7f 0.35f 1.0f 2.0f 0.7f 1.0f; do
  GFW_JIT_DEFS="GFW_P1_E_SCALE=$sc" timeout 300 python3 bench.py --gpus 1 --steps 200 --warmup 20 --no-cpu-baseline > $O/bench_$sc.json 2> $O/bench_$sc.err
  python3 -c "import json; d=json.load(open('$O/bench_$sc.json')); print('E x $sc',d['value'], d['ms_per_step'], d['roofline']['kernel_ms_per_frame'], d['config']['backend'], d['config']['parity_vs_oracle'])" 2>&1 | tail -1 | tee -a $O/summary.txt
done
python3 tools/audit_c2.py 2>&1 | tail -5 | tee -a $O/summary.txt
