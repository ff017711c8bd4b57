#!/bin/bash
# round 6, call s: C3 (8K) against the frames per launch — round 5 measured 183.9 us at 8 per launch, round 6 200.7 at 16
O=gpurun_out/r06_s; mkdir -p $O; : > $O/summary.txt
export TMPDIR=/tmp
rec() { local name="$1"; shift; timeout 600 python3 bench.py --gpus 1 "$@" > $O/bench_$name.json 2> $O/bench_$name.err
  python3 -c "import json; d=json.load(open('$O/bench_$name.json')); r=d.get('roofline', {}); print('$name [$*]', d['value'], d['ms_per_step'], r.get('kernel_ms_per_frame'), r.get('frac'), r.get('frames_per_launch'), d['config'].get('backend'), str(d['config'].get('parity_vs_oracle'))[:40])" 2>&1 | tail -1 | tee -a $O/summary.txt; tail -2 $O/bench_$name.err | grep -v amdgpu.ids | tee -a $O/summary.txt; }
C3="--no-cpu-baseline --no-parity --width 7680 --height 4320 --resident 16 --steps 96"
rec c3_clip16 $C3 --clip 16
rec c3_clip8 $C3 --clip 8
rec c3_clip4 $C3 --clip 4
rec c3_clip2 $C3 --clip 2
rec c3_clip8_res8 --no-cpu-baseline --no-parity --width 7680 --height 4320 --resident 8 --steps 96 --clip 8
rec c3_clip16_b $C3 --clip 16
rec c3_clip8_b $C3 --clip 8
rec c2_clip4 --no-cpu-baseline --no-parity --steps 200 --clip 4
rec c2_clip8 --no-cpu-baseline --no-parity --steps 200 --clip 8
rec c2_clip10 --no-cpu-baseline --no-parity --steps 200 --clip 10
