#!/bin/bash
# round 3, thirteenth GPU call: priority-feedback divisor sweep at eight waves per SIMD
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03m; mkdir -p $O
b() { name=$1; shift; timeout 400 python bench.py "$@" > $O/bench_$name.json 2> $O/bench_$name.err; python3 - <<PY
import json
try:
    d = json.load(open("$O/bench_$name.json"))
    r = d.get("roofline", {})
    print("$name", d["value"], d["ms_per_step"], r.get("kernel_ms_per_frame"), d["config"]["backend"], d["config"].get("parity_vs_oracle"))
except Exception as e:
    print("$name FAILED", e, open("$O/bench_$name.err").read()[-600:])
PY
}
A="--gpus 1 --steps 200 --warmup 20 --no-cpu-baseline"
for d in 8 12 16 24 32 64; do GFW_JIT_WAVES=8 GFW_JIT_DEFS="GFW_PRIO_DIV=$d" b c2_w8_div$d $A; done
for d in 8 16; do GFW_JIT_WAVES=7 GFW_JIT_DEFS="GFW_PRIO_DIV=$d" b c2_w7_div$d $A; done
for cfg in "nv12:--fmt NV12 --steps 200 --warmup 20" "l8:--interp 8 --steps 64 --warmup 8" "sv:--digital gopro_superview --steps 64 --warmup 8" "c3:--width 7680 --height 4320 --resident 16 --steps 48 --warmup 8" "c1:--c1 --steps 200 --warmup 20" "c4:--fmt RGBAF32 --crop --resident 16 --steps 64 --warmup 8"; do
  n=${cfg%%:*}; a=${cfg#*:}
  for d in 8 16; do GFW_JIT_WAVES=8 GFW_JIT_DEFS="GFW_PRIO_DIV=$d" b ${n}_w8_div$d --gpus 1 --no-cpu-baseline $a; done
done
GFW_JIT_WAVES=7 GFW_JIT_DEFS="GFW_PRIO_DIV=16" b c1_w7_div16 --gpus 1 --no-cpu-baseline --c1 --steps 200 --warmup 20
GFW_JIT_WAVES=7 GFW_JIT_DEFS="GFW_PRIO_DIV=16" b c4_w7_div16 --gpus 1 --no-cpu-baseline --fmt RGBAF32 --crop --resident 16 --steps 64 --warmup 8
GFW_JIT_WAVES=8 GFW_JIT_DEFS="GFW_PRIO_DIV=16" b driver_w8_div16 --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline
