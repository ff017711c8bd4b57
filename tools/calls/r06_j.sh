#!/bin/bash
# round 6, call j: reproduce (1) the generic_polynomial bench mismatch, (2) the one-pixel difference of a ptlens audit frame under four concurrent test processes
O=gpurun_out/r06_j; mkdir -p $O; : > $O/summary.txt
export TMPDIR=/tmp GFW_JIT_CACHE=/tmp/jitc; mkdir -p /tmp/jitc
rec() { local name="$1"; shift; timeout 400 python3 bench.py --gpus 1 "$@" > $O/bench_$name.json 2> $O/bench_$name.err
  python3 -c "import json; d=json.load(open('$O/bench_$name.json')); r=d.get('roofline', {}); print('$name [$*]', d['value'], d['ms_per_step'], r.get('kernel_ms_per_frame'), r.get('frac'), r.get('frames_per_launch'), d['config'].get('backend'), str(d['config'].get('parity_vs_oracle'))[:40])" 2>&1 | tail -1 | tee -a $O/summary.txt; grep "differs from" $O/bench_$name.err | tee -a $O/summary.txt; }
rec gp1 --no-cpu-baseline --lens-model generic_polynomial --steps 100
rec gp2 --no-cpu-baseline --lens-model generic_polynomial --steps 100
rec gp_clip1 --no-cpu-baseline --lens-model generic_polynomial --steps 100 --clip 1

rec sony --no-cpu-baseline --lens-model sony --steps 100
mkdir -p $O/jitc; cp /tmp/jitc/* $O/jitc/
unset GFW_JIT_CACHE
for i in 1 2 3; do
  timeout 900 python3 -m pytest tests/test_gpu_pass1_radial.py -q -m gpu --tb=line -p no:cacheprovider -n 4 2>&1 | grep -v "amdgpu.ids" | tail -12 | tee -a $O/summary.txt
done
