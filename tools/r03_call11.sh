#!/bin/bash
# round 3, eleventh GPU call: eight waves per SIMD for the baked bilinear kernel (it shrank: 68 VGPRs at 7) — with and without the priority feedback
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03k; mkdir -p $O
b() { name=$1; shift; timeout 400 python bench.py "$@" > $O/bench_$name.json 2> $O/bench_$name.err; python3 - <<PY
import json
try:
    d = json.load(open("$O/bench_$name.json"))
    r = d.get("roofline", {})
    print("$name", d["value"], d["ms_per_step"], r.get("kernel_ms_per_frame"), r.get("frames_per_launch"), d["config"]["backend"], d["config"].get("parity_vs_oracle"), d["config"].get("jit", {}).get("compile_ms"), d["config"].get("host_enqueue_ms_per_step"))
except Exception as e:
    print("$name FAILED", e, open("$O/bench_$name.err").read()[-600:])
PY
}
A="--gpus 1 --steps 200 --warmup 20 --no-cpu-baseline"
b base $A
GFW_JIT_WAVES=8 b w8 $A
GFW_JIT_WAVES=8 GFW_JIT_DEFS="GFW_PRIO_MODE=0" b w8_prio0 $A
GFW_JIT_WAVES=8 GFW_JIT_DEFS="GFW_PRIO_MODE=0;GFW_P3_SPLIT=1" b w8_prio0_split $A
GFW_JIT_WAVES=8 GFW_JIT_DEFS="GFW_PRIO_DIV=4" b w8_div4 $A
GFW_JIT_WAVES=8 GFW_JIT_DEFS="GFW_PRIO_MODE=0" b w8_prio0_b $A
GFW_JIT_WAVES=8 GFW_JIT_DEFS="GFW_PRIO_MODE=0" b w8_prio0_g1792 $A --grid 1792
GFW_JIT_WAVES=8 GFW_JIT_DEFS="GFW_PRIO_MODE=0" b w8_prio0_c1 $A --c1
b base_c1 $A --c1
GFW_JIT_WAVES=8 GFW_JIT_DEFS="GFW_PRIO_MODE=0" b w8_prio0_c3 --gpus 1 --steps 48 --warmup 8 --no-cpu-baseline --width 7680 --height 4320 --resident 16
GFW_JIT_WAVES=8 GFW_JIT_DEFS="GFW_PRIO_MODE=0" b w8_prio0_lanczos --gpus 1 --steps 64 --warmup 8 --no-cpu-baseline --interp 8
GFW_JIT_WAVES=8 GFW_JIT_DEFS="GFW_PRIO_MODE=0" b w8_prio0_bicubic --gpus 1 --steps 64 --warmup 8 --no-cpu-baseline --interp 4
GFW_JIT_WAVES=8 GFW_JIT_DEFS="GFW_PRIO_MODE=0" b w8_prio0_c4 --gpus 1 --steps 64 --warmup 8 --no-cpu-baseline --fmt RGBAF32 --crop --resident 16
GFW_JIT_WAVES=8 GFW_JIT_DEFS="GFW_PRIO_MODE=0" b w8_prio0_superview --gpus 1 --steps 64 --warmup 8 --no-cpu-baseline --digital gopro_superview
b base_b $A
