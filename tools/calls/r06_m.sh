#!/bin/bash
# round 6, call m: where EWA stands (per-plane kernel): C2 and 1080p NV12, interpolation 10 (the family's first member)
O=gpurun_out/r06_m; mkdir -p $O; : > $O/summary.txt
export TMPDIR=/tmp
rec() { local name="$1"; shift; timeout 600 python3 bench.py --gpus 1 "$@" > $O/bench_$name.json 2> $O/bench_$name.err
  python3 -c "import json; d=json.load(open('$O/bench_$name.json')); r=d.get('roofline', {}); print('$name [$*]', d['value'], d['ms_per_step'], r.get('kernel_ms_per_frame'), r.get('frac'), r.get('frames_per_launch'), d['config'].get('backend'), str(d['config'].get('parity_vs_oracle'))[:40])" 2>&1 | tail -1 | tee -a $O/summary.txt; tail -2 $O/bench_$name.err | grep -v amdgpu.ids | tee -a $O/summary.txt; }
rec c2_ewa10 --no-cpu-baseline --no-parity --interp 10 --steps 10 --warmup 2
rec c1_ewa10 --no-cpu-baseline --no-parity --interp 10 --steps 10 --warmup 2 --c1
rec nv12_ewa12 --no-cpu-baseline --no-parity --interp 12 --steps 10 --warmup 2 --fmt NV12
