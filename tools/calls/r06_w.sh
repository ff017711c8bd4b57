#!/bin/bash
# round 6, call w: the EWA tap loop with the dead taps leaving before their root (dr >= 4): parity tests that reach EWA, then the benches of call m
O=gpurun_out/r06_w; mkdir -p $O; : > $O/summary.txt
export TMPDIR=/tmp
timeout 1500 python3 -m pytest tests/test_gpu_fuzz.py tests/test_gpu_parity.py tests/test_gpu_fused_coverage.py -q -m gpu --tb=short -p no:cacheprovider -n 4 2>&1 | grep -v "amdgpu.ids" | tail -4 | tee -a $O/summary.txt
rec() { local name="$1"; shift; timeout 600 python3 bench.py --gpus 1 "$@" > $O/bench_$name.json 2> $O/bench_$name.err
  python3 -c "import json; d=json.load(open('$O/bench_$name.json')); r=d.get('roofline', {}); print('$name [$*]', d['value'], d['ms_per_step'], r.get('kernel_ms_per_frame'), r.get('frac'), r.get('frames_per_launch'), d['config'].get('backend'), str(d['config'].get('parity_vs_oracle'))[:40])" 2>&1 | tail -1 | tee -a $O/summary.txt; tail -2 $O/bench_$name.err | grep -v amdgpu.ids | tee -a $O/summary.txt; }
rec c2_ewa10 --no-cpu-baseline --no-parity --interp 10 --steps 10 --warmup 2
rec c1_ewa10 --no-cpu-baseline --interp 10 --steps 10 --warmup 2 --c1
rec nv12_ewa12 --no-cpu-baseline --no-parity --interp 12 --steps 10 --warmup 2 --fmt NV12
rec rgba_ewa10 --no-cpu-baseline --no-parity --interp 10 --steps 10 --warmup 2 --fmt RGBA --width 1920 --height 1080
