#!/bin/bash
# round 3, twelfth GPU call: eight waves per SIMD with the priority feedback (divisor sweep) across the configurations
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03l; mkdir -p $O
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
S="--gpus 1 --steps 64 --warmup 8 --no-cpu-baseline"
for d in 4 5 6 8; do GFW_JIT_WAVES=8 GFW_JIT_DEFS="GFW_PRIO_DIV=$d" b c2_w8_div$d $A; done
GFW_JIT_WAVES=7 GFW_JIT_DEFS="GFW_PRIO_DIV=4" b c2_w7_div4 $A
for cfg in "c1:--c1 --steps 200 --warmup 20" "c4:--fmt RGBAF32 --crop --resident 16 --steps 64 --warmup 8" "sv:--digital gopro_superview --steps 64 --warmup 8" "l8:--interp 8 --steps 64 --warmup 8" "l4:--interp 4 --steps 64 --warmup 8" "nv12:--fmt NV12 --steps 200 --warmup 20" "p010:--fmt P010 --steps 200 --warmup 20" "c4p:--fmt GBRAPF32LE --crop --resident 16 --steps 64 --warmup 8"; do
  n=${cfg%%:*}; a=${cfg#*:}
  b ${n}_w7 --gpus 1 --no-cpu-baseline $a
  GFW_JIT_WAVES=8 GFW_JIT_DEFS="GFW_PRIO_DIV=4" b ${n}_w8_div4 --gpus 1 --no-cpu-baseline $a
done
