#!/bin/bash
# round 6, call ad: rows per lane of the specialised kernel (GFW_JIT_RB_FAST: tile height = 4 x RB) against the launch size — does a frame-by-frame caller want smaller tiles?
O=gpurun_out/r06_ad; mkdir -p $O; : > $O/summary.txt
export TMPDIR=/tmp
rec() { local name="$1"; shift; timeout 600 python3 bench.py --gpus 1 "$@" > $O/bench_$name.json 2> $O/bench_$name.err
  python3 -c "import json; d=json.load(open('$O/bench_$name.json')); r=d.get('roofline', {}); print('$name [$*]', d['value'], d['ms_per_step'], r.get('kernel_ms_per_frame'), r.get('frac'), r.get('frames_per_launch'), d['config'].get('backend'), str(d['config'].get('parity_vs_oracle'))[:40])" 2>&1 | tail -1 | tee -a $O/summary.txt; tail -2 $O/bench_$name.err | grep -v amdgpu.ids | tee -a $O/summary.txt; }
for rb in 1 2 3 4 6 8; do
  GFW_JIT_RB_FAST=$rb GFW_JIT_CACHE=/tmp/jitc_rb rec c2_clip1_rb$rb --no-cpu-baseline --steps 200 --clip 1
done
for rb in 2 3 4 6 8; do
  GFW_JIT_RB_FAST=$rb GFW_JIT_CACHE=/tmp/jitc_rb rec c2_clip10_rb$rb --no-cpu-baseline --steps 200
done
