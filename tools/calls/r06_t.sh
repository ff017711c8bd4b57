#!/bin/bash
# round 6, call t: is C3's 184 / 200 us a property of the process (where its buffers landed) or of the frames per launch?  The same command eight times, addresses printed
O=gpurun_out/r06_t; mkdir -p $O; : > $O/summary.txt
export TMPDIR=/tmp GFW_BENCH_ADDR=1
rec() { local name="$1"; shift; timeout 600 python3 bench.py --gpus 1 "$@" > $O/bench_$name.json 2> $O/bench_$name.err
  python3 -c "import json; d=json.load(open('$O/bench_$name.json')); r=d.get('roofline', {}); print('$name [$*]', d['value'], d['ms_per_step'], r.get('kernel_ms_per_frame'), r.get('frac'), r.get('frames_per_launch'))" 2>&1 | tail -1 | tee -a $O/summary.txt; grep "^addr" $O/bench_$name.err | tee -a $O/summary.txt; }
C3="--no-cpu-baseline --no-parity --width 7680 --height 4320 --resident 16 --steps 96"
for i in 1 2 3 4; do rec c3_clip8_$i $C3 --clip 8; rec c3_clip16_$i $C3 --clip 16; done
rocm-smi --showclocks --showpower 2>/dev/null | grep -v "^=\|^$" | head -12 | tee -a $O/summary.txt
